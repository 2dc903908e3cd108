#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call8.txt; : > $O
for i in 1 2 3; do python -m pytest tests/test_optim_gpu.py -q -x 2>&1 | grep -E "^E  .*(AssertionError|assert|\(')|passed|failed" | cut -c1-400 | head -8 >> $O; done
cat $O
