#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call17.txt; : > $O
python tools/abl/ln_rows_bench.py >> $O 2>&1
python -m pytest tests/test_models_gpu.py -q -x --tb=short -k "pkgm" 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-300 | tail -4 >> $O
for v in 1 0; do
  IA_LN_ROWS=$v TAG=r06tmp bash tools/runs/run.sh prof lnrows$v python3 bench.py --single-stream --steps 3 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > /dev/null 2>&1
  echo "IA_LN_ROWS=$v: $(grep -E 'ln_bwd_kernel' gpurun_out/r06tmp_lnrows${v}_kernel_stats_summary.txt)" >> $O
done
for v in 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v python tools/config_bench.py c4 2>&1 | grep -E "pairs/s" >> $O; done
cat $O
