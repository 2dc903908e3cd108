#!/bin/bash
# kernel-trace stats of the default bench workload (5 steps) -> gpurun_out/quick_summary.txt
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktq
rocprofv3 --kernel-trace --stats -d /tmp/ktq -o b --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > $R/gpurun_out/quick_bench.json 2> /dev/null
python3 $R/tools/prof_summary.py /tmp/ktq/b_kernel_stats.csv 7 45 > $R/gpurun_out/quick_summary.txt
