#!/bin/bash
python -m pytest tests/test_baseline_shapes_gpu.py::test_c2_full_width_one_tower tests/test_models_gpu.py tests/test_kernels_gpu.py::test_gemm_dynamic_tile_claim_under_cu_contention tests/test_engine_gpu.py "tests/test_kernels_gpu.py::test_gemm_nt_epilogues" "tests/test_kernels_gpu.py::test_gemm_persistent_rounds_with_clipped_tiles" -q 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | cut -c1-900 | head -40 > gpurun_out/r05_t5.log
cat gpurun_out/r05_t5.log
python tools/abl/dyn_ab.py > gpurun_out/dyn_ab.txt 2>&1; tail -8 gpurun_out/dyn_ab.txt
for v in 0 1 0 1; do IA_GEMM_DYNAMIC=$v python bench.py --no-pmc --no-cpu-baseline --no-variants --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('IA_GEMM_DYNAMIC=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms')"; done | tee gpurun_out/ab_dynamic_step.txt
for v in 0 1; do IA_GEMM_DYNAMIC=$v python bench.py --no-pmc --no-cpu-baseline --no-variants --steps 20 --warmup 5 --pairs-per-gpu 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('16 pairs/GPU IA_GEMM_DYNAMIC=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],2), 'ms')"; done | tee -a gpurun_out/ab_dynamic_step.txt
cat gpurun_out/c2_full_width_gradients.txt
python tools/parity_report.py 2>/dev/null | grep "two_tower_hinge" | head -20
