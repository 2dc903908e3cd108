#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call4.txt; : > $O
python -m pytest tests/test_models_gpu.py -k "coca" -q 2>&1 | tail -3 >> $O
for n in 16 32 64 96; do IA_CB_PAIRS=$n python tools/config_bench.py c5x 2>&1 | grep -E "pairs/s|Error|error" >> $O; done
IA_DROPOUT_SEEDS=256 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed" >> $O
IA_DROPOUT_SEEDS=64 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed" >> $O
cat $O
