#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call7.txt; : > $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -12 >> $O
for v in 1 0 1 0; do echo "IA_NFNET_FUSE_TAIL=$v (32 pairs)" >> $O; IA_CB_PAIRS=32 IA_NFNET_FUSE_TAIL=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O
bash tools/runs/run.sh quick >> $O 2>&1
cat $O
