#!/bin/bash
R=$GRAFT_REPO_ROOT
for d in 0 64 32 2 34 16 50; do
echo "== IA_GEMM_DBG=$d"
IA_GEMM_DBG=$d python3 $R/tools/abl/gemm_ksweep.py 65280 1024 2>&1 | grep -v amdgpu.ids | grep "K=   64\|K=  256\|K= 1024\|overhead"
done
