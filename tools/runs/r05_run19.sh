#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in c4 c5x; do
  rm -rf /tmp/kt_$cfg
  rocprofv3 --kernel-trace --stats -d /tmp/kt_$cfg -o b --output-format csv -- python3 $R/tools/config_bench.py $cfg > /tmp/${cfg}_log.txt 2>&1
  echo "== $cfg"; python3 $R/tools/prof_summary.py $(find /tmp/kt_$cfg -name "*kernel_stats.csv" | head -1) 11 26
done
