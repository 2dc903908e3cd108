#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
set1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
cp $R/item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
run() { rocprofv3 --kernel-trace --pmc $set1 -d $R/gpurun_out/vpmc_$1 -o p -- python3 $R/tools/abl/gemm_vs_vendor_pmc.py $2 $3 $4 > $R/gpurun_out/vpmc_$1.log 2>&1; echo "$1 rc=$?"; }
for v in "$@"; do
  lib=${v%%:*}; dbg=${v##*:}
  if [ "$lib" != default ]; then cp $R/tools/abl/lib_$lib.so $R/item_alignment_amd/libitemalign_hip.so; fi
  IA_GEMM_DBG=$dbg run ${lib}_dbg$dbg 65280 4096 2048
  cp /tmp/lib_default.so $R/item_alignment_amd/libitemalign_hip.so
done
