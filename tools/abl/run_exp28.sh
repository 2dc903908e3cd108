#!/bin/bash
R=$GRAFT_REPO_ROOT
for d in 0 4 36 2; do
echo "== IA_GEMM_DBG=$d"
IA_GEMM_DBG=$d python3 $R/tools/abl/gemm_ksweep.py 65280 4096 2>&1 | grep -v amdgpu.ids | grep "K= 1024\|K= 4096\|overhead"
done
