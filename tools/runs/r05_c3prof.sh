#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_c3
IA_CB_PAIRS=32 rocprofv3 --kernel-trace --stats -d /tmp/kt_c3 -o b --output-format csv -- python3 $R/tools/config_bench.py c3 > /tmp/c3_log.txt 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/kt_c3 -name "*kernel_stats.csv" | head -1) 11 40 > $R/gpurun_out/r05_c3_32_kernel_stats_summary.txt
tail -1 /tmp/c3_log.txt >> $R/gpurun_out/r05_c3_32_kernel_stats_summary.txt
cat $R/gpurun_out/r05_c3_32_kernel_stats_summary.txt
