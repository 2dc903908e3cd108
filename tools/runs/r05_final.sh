#!/bin/bash
# round-5 wrap-up on one box: the whole GPU suite, then every profile DESIGN.md §7 cites (copied to profiles/ afterwards)
mkdir -p gpurun_out/profiles
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r05_gpu_suite.txt; cat gpurun_out/r05_gpu_suite.txt
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect.log 2>&1
python tools/config_bench.py c2 c4 c5x 2>&1 | grep -E "pairs/s" > gpurun_out/profiles/r05_config_bench_other.txt
for n in 16 32 64; do IA_CB_PAIRS=$n python tools/config_bench.py c3 2>&1 | grep -E "pairs/s"; done > gpurun_out/profiles/r05_c3_batch_sweep.txt
IA_CB_PAIRS=32 IA_CONV_DIRECT=0 python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" | sed 's/^/IA_CONV_DIRECT=0 (round-4 GEMM-view path): /' >> gpurun_out/profiles/r05_c3_batch_sweep.txt
IA_CB_PAIRS=32 IA_CONV_DIRECT_WGRAD=0 python tools/config_bench.py c3 2>&1 | grep -E "pairs/s" | sed 's/^/IA_CONV_DIRECT_WGRAD=0 (direct forward + data gradient only): /' >> gpurun_out/profiles/r05_c3_batch_sweep.txt
cat gpurun_out/profiles/r05_config_bench_other.txt gpurun_out/profiles/r05_c3_batch_sweep.txt
python tools/conv_small_bench.py 32 > gpurun_out/profiles/r05_conv_small_bench.txt 2>&1
python bench.py > gpurun_out/profiles/r05_bench_line.json 2> gpurun_out/r05_bench_line.err
python -c "
import json; d=json.load(open('gpurun_out/profiles/r05_bench_line.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic'), d['mfma_fraction_whole_step'], d['mfma_fraction_dense'])
print({k:round(v['value'],1) for k,v in d['variants'].items()})
print(d.get('cpu_baseline'))"
# verdict 2(d): the same kernels with q pre-scaled by the QKV projection (default) and scaled inside the attention kernels, same box
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ps in 1 0; do
  rm -rf /tmp/kt_ps$ps
  IA_Q_PRESCALE=$ps rocprofv3 --kernel-trace --stats -d /tmp/kt_ps$ps -o b --output-format csv -- python3 $R/bench.py --single-stream --steps 4 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > /tmp/ps$ps.json 2>/dev/null
  echo "IA_Q_PRESCALE=$ps: $(python3 -c "import json;d=json.load(open('/tmp/ps$ps.json'));print(round(d['value'],1),'pairs/s')")"
  python3 $R/tools/prof_summary.py $(find /tmp/kt_ps$ps -name "*kernel_stats.csv" | head -1) 6 40 | grep -E "attn|total"
done > $R/gpurun_out/profiles/r05_ab_q_prescale_per_kernel.txt 2>&1
cat $R/gpurun_out/profiles/r05_ab_q_prescale_per_kernel.txt
