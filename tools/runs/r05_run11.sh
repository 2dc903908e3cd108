#!/bin/bash
# the whole GPU suite (summary lines kept), conv_small_bench with the direct data gradient, two-stream headline static against dynamic tile order, the bench line
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E  |^FAILED|flash-style" | cut -c1-300 > gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_gpu_suite.txt
python tools/conv_small_bench.py 32 > gpurun_out/profiles/r05_conv_small_bench.txt 2>&1; cat gpurun_out/profiles/r05_conv_small_bench.txt
for d in 0 1 0 1; do
  IA_GEMM_DYNAMIC=$d python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-pmc --no-variants 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('two streams, IA_GEMM_DYNAMIC=$d:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms')"
done | tee gpurun_out/profiles/r05_ab_two_stream_tile_order.txt
python bench.py > gpurun_out/profiles/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'])"
