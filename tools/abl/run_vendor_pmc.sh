#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_WAVE32_LDS SQ_WAVES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/vpmc$i -o p -- python3 $R/tools/abl/gemm_vs_vendor_pmc.py > $R/gpurun_out/vpmc$i.log 2>&1
  echo "set $i rc=$?"
done
