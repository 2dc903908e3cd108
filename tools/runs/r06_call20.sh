#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call20.txt; : > $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -k "lookahead or gemm" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tail -6 >> $O
timeout 600 python tools/abl/gemm_la_ab.py 2>&1 | grep -v amdgpu.ids >> $O
for v in 1 0 1 0; do echo "IA_GEMM_LA=$v: $(IA_GEMM_LA=$v timeout 600 bash tools/runs/run.sh quick 2>&1 | tail -1)" >> $O; done
cat $O
