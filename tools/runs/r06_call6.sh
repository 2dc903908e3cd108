#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call6.txt; : > $O
python -m pytest tests/test_kernels_gpu.py -k "exact_delta" -q -x -s 2>&1 | grep -E "adverse|passed|failed|Error|assert" | cut -c1-400 >> $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 >> $O
cat $O
