#!/bin/bash
cd $GRAFT_REPO_ROOT
IA_DROPOUT_SEEDS=256 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed"
IA_DROPOUT_SEEDS=64 python -m pytest tests/test_models_gpu.py -k dropout_on -s -q 2>&1 | grep -E "dropout z|passed|failed"
python tools/c5x_mem_probe.py 16 2>&1 | grep -v "Warning\|amdgpu.ids\|getattr" | tee gpurun_out/r06_c5x_mem_probe16.txt
PROF_STEPS=11 bash tools/runs/run.sh prof c5x python3 tools/config_bench.py c5x
