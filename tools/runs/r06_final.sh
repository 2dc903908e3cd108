#!/bin/bash
# round-6 wrap-up on one box: every profile DESIGN.md 7 cites (copied to profiles/ afterwards)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/profiles
bash tools/collect_profiles.sh r06 > gpurun_out/r06_collect.log 2>&1
python tools/config_bench.py c2 c4 c5x 2>&1 | grep -E "pairs/s" > gpurun_out/profiles/r06_config_bench_other.txt
python tools/config_bench.py c3r c3b c3b3 2>&1 | grep -E "pairs/s" > gpurun_out/profiles/r06_config_bench_resnets.txt
cat gpurun_out/profiles/r06_config_bench_other.txt gpurun_out/profiles/r06_config_bench_resnets.txt
python tools/parity_report.py > gpurun_out/profiles/r06_parity_report.txt 2>&1
grep -E "passed|failed|pytest exit|gradient cosine:|output relative" gpurun_out/profiles/r06_parity_report.txt
python bench.py > gpurun_out/profiles/r06_bench_line.json 2> gpurun_out/r06_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r06_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'])
print({k:round(v['value'],1) for k,v in d['variants'].items()})
print(d.get('attention_text'))
print(d.get('cpu_baseline'))"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/profiles/r06_bench_driver_flags.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/profiles/r06_bench_driver_flags.json')); print('driver flags:', d['value'], d['ms_per_step'], d['roofline']['frac'])"
ls gpurun_out/profiles | grep r06
