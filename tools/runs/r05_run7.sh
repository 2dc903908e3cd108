#!/bin/bash
python -m pytest tests/test_baseline_shapes_gpu.py::test_c2_full_width_one_tower "tests/test_models_gpu.py::test_roberta_two_tower" tests/test_kernels_gpu.py::test_gemm_dynamic_tile_claim_under_cu_contention tests/test_dp_gpu.py tests/test_optim_gpu.py -q 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | cut -c1-400 | head -30 > gpurun_out/r05_t7.log
cat gpurun_out/r05_t7.log
python tools/abl/dyn_ab.py > gpurun_out/dyn_ab.txt 2>&1; tail -8 gpurun_out/dyn_ab.txt
python tools/cu_contention.py > gpurun_out/r05_cu_contention.txt 2>&1; tail -16 gpurun_out/r05_cu_contention.txt
for v in 0 1; do IA_GEMM_DYNAMIC=$v python bench.py --no-pmc --no-cpu-baseline --no-variants --steps 20 --warmup 5 --pairs-per-gpu 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('16 pairs/GPU IA_GEMM_DYNAMIC=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms')"; done | tee gpurun_out/ab_dynamic_step16.txt
