#!/bin/bash
# round-5 profile collection: bench kernel stats + PMC (tools/collect_profiles.sh), the other configurations, the C3 batch sweep
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
python tools/config_bench.py c2 c4 c5x 2>&1 | grep -E "pairs/s" > gpurun_out/profiles/r05_config_bench_other.txt
for n in 16 24 32 48; do IA_CB_PAIRS=$n python tools/config_bench.py c3 2>&1 | grep -E "pairs/s"; done > gpurun_out/profiles/r05_c3_batch_sweep.txt
cat gpurun_out/profiles/r05_config_bench_other.txt gpurun_out/profiles/r05_c3_batch_sweep.txt
python bench.py > gpurun_out/profiles/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'])
print({k:round(v['value'],1) for k,v in d['variants'].items()})
print(d.get('cpu_baseline'))"
