#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call11.txt; : > $O
python -m pytest tests/test_kernels_gpu.py -k "skips_query or eca_block or attention_fwd_bwd or bwd_bias or dropout_mask or bench_shapes or first_launch" -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed" | cut -c1-400 | tail -6 >> $O
python -m pytest tests/test_models_gpu.py tests/test_engine_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
python -m pytest tests/test_baseline_shapes_gpu.py -q -x --tb=short 2>&1 | grep -E "^E  |passed|failed|^FAILED" | cut -c1-400 | tail -6 >> $O
for v in 1 0 1 0; do echo "IA_MASKED_ROWS_DEAD=$v" >> $O; IA_MASKED_ROWS_DEAD=$v bash tools/runs/run.sh quick >> $O 2>&1; done
for v in 1 2 1 2; do echo "IA_NFNET_FUSE_TAIL=$v (32 pairs)" >> $O; IA_CB_PAIRS=32 IA_NFNET_FUSE_TAIL=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
cat $O
