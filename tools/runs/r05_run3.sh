#!/bin/bash
timeout 120 python tools/abl/dyn_debug.py 2>&1 | tail -12
python -m pytest tests/test_baseline_shapes_gpu.py::test_c2_full_width_one_tower tests/test_models_gpu.py tests/test_optim_gpu.py tests/test_kernels_gpu.py::test_gemm_dynamic_tile_claim_under_cu_contention -q 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | head -80 > gpurun_out/r05_t3.log
cat gpurun_out/r05_t3.log
IA_GEMM_DYNAMIC=1 python tools/cu_contention.py > gpurun_out/cu_contention_dynamic.txt 2>&1
tail -9 gpurun_out/cu_contention_dynamic.txt
python tools/abl/nn_vs_nt.py > gpurun_out/nn_vs_nt.txt 2>&1; tail -10 gpurun_out/nn_vs_nt.txt
