#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/wg_pmc -o p --output-format csv -- python3 $R/tools/abl/wgrad_time.py > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/wg_pmc/*counter_collection.csv | head -1) | grep -E "t256w|kernel,"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/dx_pmc -o p --output-format csv -- python3 $R/tools/kbench.py gemm > /dev/null 2>&1
python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/dx_pmc/*counter_collection.csv | head -1) | grep -E "t256w"
