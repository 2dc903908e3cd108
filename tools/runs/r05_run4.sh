#!/bin/bash
python -m pytest tests/test_baseline_shapes_gpu.py::test_c2_full_width_one_tower tests/test_models_gpu.py tests/test_optim_gpu.py tests/test_kernels_gpu.py::test_gemm_dynamic_tile_claim_under_cu_contention tests/test_engine_gpu.py tests/test_dp_gpu.py -q 2>&1 | grep "^E  \|^FAILED\|passed\|failed\|Error" | cut -c1-600 | head -60 > gpurun_out/r05_t4.log
cat gpurun_out/r05_t4.log
python tools/abl/dyn_ab.py > gpurun_out/dyn_ab.txt 2>&1; tail -8 gpurun_out/dyn_ab.txt
for v in 0 1; do IA_DGRAD_NT=$v python bench.py --no-pmc --no-cpu-baseline --no-variants --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('IA_DGRAD_NT=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms', 'frac', round(d['roofline']['frac'],4))"; done | tee gpurun_out/ab_dgrad_nt.txt
for v in 0 1 0 1; do IA_GEMM_DYNAMIC=$v python bench.py --no-pmc --no-cpu-baseline --no-variants --steps 8 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('IA_GEMM_DYNAMIC=$v', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms')"; done | tee gpurun_out/ab_dynamic_step.txt
python tools/parity_report.py 2>&1 | grep -i "hinge" | head -20
