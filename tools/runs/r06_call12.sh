#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call12.txt; : > $O
python -m pytest tests/test_models_gpu.py tests/test_engine_gpu.py tests/test_cli_gpu.py tests/test_dp_gpu.py tests/test_optim_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -8 >> $O
cat $O
