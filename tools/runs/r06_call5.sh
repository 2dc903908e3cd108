#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_call5.txt; : > $O
python -m pytest tests/test_kernels_gpu.py -k "eca or conv3x3" -q -x 2>&1 | tail -4 >> $O
python -m pytest tests/test_models_gpu.py -k "nfnet or dropout_on or resnet" -q -x 2>&1 | tail -4 >> $O
python -m pytest tests/test_baseline_shapes_gpu.py -k "c3 or c5" -q -x 2>&1 | tail -4 >> $O
python -m pytest tests/test_optim_gpu.py -q -x 2>&1 | tail -3 >> $O
for v in 1 0 1 0; do echo "IA_ECA_LINEAR=$v" >> $O; IA_ECA_LINEAR=$v python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" >> $O; done
PROF_STEPS=11 bash tools/runs/run.sh prof c5x64 python3 tools/config_bench.py c5x > /dev/null 2>&1
PROF_STEPS=11 bash tools/runs/run.sh prof c3 python3 tools/config_bench.py c3 > /dev/null 2>&1
cat $O
