#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call13.txt; : > $O
python -m pytest tests/test_kernels_gpu.py -k "skips_query or attention" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
bash tools/runs/run.sh suite >> $O 2>&1
for v in 1 0 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v python tools/config_bench.py c2 2>&1 | grep -E "pairs/s" >> $O; done
for v in 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v bash tools/runs/run.sh quick >> $O 2>&1; done
cat $O
