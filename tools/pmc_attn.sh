#!/bin/bash
# usage (on the GPU box): tools/pmc_attn.sh <binary> <args...> ; prints per-kernel PMC sums (separate passes per counter set)
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; BIN=$1; shift
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS"; do
  i=$((i+1)); rocprofv3 --pmc $set -d $R/gpurun_out/pmcx_$i -o x --output-format csv -- $BIN "$@" > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/pmcx_$i/*counter_collection.csv | head -1) | grep -v "^kernel"
done
