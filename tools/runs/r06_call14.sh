#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call14.txt; : > $O
python -m pytest tests/test_engine_gpu.py tests/test_models_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
for v in 1 0 1 0; do echo "IA_LN_ROWS=$v: $(IA_LN_ROWS=$v bash tools/runs/run.sh quick 2>&1 | tail -1)" >> $O; done
cat $O
