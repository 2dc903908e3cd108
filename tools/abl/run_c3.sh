#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests -q -x -m gpu -k "nfnet or c3 or resnet" 2>&1 | tail -3
rm -rf /tmp/kt_c3
rocprofv3 --kernel-trace --stats -d /tmp/kt_c3 -o b --output-format csv -- python3 $R/tools/config_bench.py c3 > /tmp/c3_log.txt 2>&1
tail -2 /tmp/c3_log.txt | grep -v rocprofv3
python3 $R/tools/prof_summary.py /tmp/kt_c3/b_kernel_stats.csv 11 24
