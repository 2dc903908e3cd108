#!/bin/bash
mkdir -p gpurun_out/profiles
python -m pytest tests/test_baseline_shapes_gpu.py -q -k "c4" 2>&1 | grep -E "passed|failed|^E  " | cut -c1-400; cat gpurun_out/c4_table_gradients.txt
python tools/conv_small_bench.py 32 > gpurun_out/profiles/r05_conv_small_bench.txt 2>&1; cat gpurun_out/profiles/r05_conv_small_bench.txt
python bench.py > gpurun_out/profiles/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['roofline'].get('traffic_source'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'])
print({k:round(v['value'],1) for k,v in d['variants'].items()})
print(d.get('cpu_baseline'))"
