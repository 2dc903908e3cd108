#!/bin/bash
# L2 hit rate and memory-side requests of our GEMM and the vendor's (two shapes)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { rocprofv3 --kernel-trace --pmc $2 -d $R/gpurun_out/vtcc_$1 -o p -- python3 $R/tools/abl/gemm_vs_vendor_pmc.py $3 $4 $5 > $R/gpurun_out/vtcc_$1.log 2>&1; echo "$1 rc=$?"; }
run ffn2_a "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" 65280 1024 4096
run ffn2_b "TCC_EA0_WRREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 65280 1024 4096
run k2048_a "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE" 65280 4096 2048
run k2048_b "TCC_EA0_WRREQ_sum TCC_REQ_sum GRBM_GUI_ACTIVE" 65280 4096 2048
