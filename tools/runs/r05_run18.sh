#!/bin/bash
python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -q -k "maxpool_stem or resnet" 2>&1 | grep -E "passed|failed|^E  |^FAILED" | cut -c1-300
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kt_c3r
rocprofv3 --kernel-trace --stats -d /tmp/kt_c3r -o b --output-format csv -- python3 $R/tools/config_bench.py c3r > /tmp/c3r_log.txt 2>&1
python3 $R/tools/prof_summary.py $(find /tmp/kt_c3r -name "*kernel_stats.csv" | head -1) 11 60 | grep -E "total|patches"
tail -1 /tmp/c3r_log.txt | cut -c1-200
