#!/bin/bash
# round-5 GPU call 2: the failing tests in detail, the CU-contention table (static vs dynamic tile order), the PMC child of bench.py
python -m pytest tests/test_baseline_shapes_gpu.py::test_c2_full_width_one_tower tests/test_models_gpu.py tests/test_optim_gpu.py tests/test_kernels_gpu.py::test_gemm_dynamic_tile_claim_under_cu_contention -q 2>&1 | grep "^E  \|^FAILED\|passed\|failed" | head -80 > gpurun_out/r05_t2b.log
IA_GEMM_DYNAMIC=0 python tools/cu_contention.py > gpurun_out/cu_contention_static.txt 2>&1
IA_GEMM_DYNAMIC=1 python tools/cu_contention.py > gpurun_out/cu_contention_dynamic.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d /tmp/pp -o p --output-format csv -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-pmc --no-variants --resident-batches 2 --single-stream > gpurun_out/pmc_child.log 2>&1
tail -5 gpurun_out/pmc_child.log
cat gpurun_out/r05_t2b.log | tail -30
