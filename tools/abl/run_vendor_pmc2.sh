#!/bin/bash
# clocks and wave cycles of build variants of our GEMM (same counters as run_vendor_pmc.sh set 1)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
set1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
cp $R/item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
run() { rocprofv3 --kernel-trace --pmc $set1 -d $R/gpurun_out/vpmc_$1 -o p -- python3 $R/tools/abl/gemm_vs_vendor_pmc.py $2 $3 $4 > $R/gpurun_out/vpmc_$1.log 2>&1; echo "$1 rc=$?"; }
run default 65280 4096 2048
run default_ffn2 65280 1024 4096
cp $R/tools/abl/lib_m16.so $R/item_alignment_amd/libitemalign_hip.so
run m16 65280 4096 2048
cp /tmp/lib_default.so $R/item_alignment_amd/libitemalign_hip.so
export IA_GEMM_DBG=16
run nobarrier 65280 4096 2048
export IA_GEMM_DBG=2
run nodma 65280 4096 2048
export IA_GEMM_DBG=4
run l2hot 65280 4096 2048
