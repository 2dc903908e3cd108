#!/bin/bash
# final collection on the final code: kernel stats + PMC (collect_profiles.sh), the bench line, the other configurations
mkdir -p gpurun_out/profiles
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
python tools/config_bench.py c2 c4 c5x 2>&1 | grep -E "pairs/s" > gpurun_out/profiles/r05_config_bench_other.txt
python bench.py > gpurun_out/profiles/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'], d['peak_hbm_gib'])
print({k:round(v['value'],1) for k,v in d['variants'].items()})
print(d.get('cpu_baseline'))"
cat gpurun_out/profiles/r05_config_bench_other.txt gpurun_out/profiles/r05_c3_hbm.txt gpurun_out/profiles/r05_c3r_hbm.txt
head -3 gpurun_out/profiles/r05_bench_kernel_stats_summary.txt
