#!/bin/bash
# rocprofv3 kernel stats of a short bench run -> gpurun_out/<tag>_kernel_stats.csv + summary.  usage: tools/prof_bench.sh <tag> [bench args]
# (environment variables such as IA_GEMM_WIDE are inherited by the profiled python3 process)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o $tag --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 4 --warmup 3 "$@" > $out/${tag}_bench.json 2> $out/${tag}_bench.err
f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
cp $f $out/${tag}_kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/prof_summary.py $f 7 40 > $out/${tag}_summary.txt
cat $out/${tag}_summary.txt
