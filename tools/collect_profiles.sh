#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench workload plus the PMC passes behind
# bench.py's roofline object (separate --pmc passes, as the guide prescribes).  Output: gpurun_out/profiles/<tag>_*.
# usage: tools/collect_profiles.sh <tag>      e.g. r02
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (--single-stream: per-kernel times of kernels that run alone; the headline runs the towers on two streams, where concurrent kernels stretch each other)
rocprofv3 --kernel-trace --stats -d $OUT/kt -o b --output-format csv -- python3 $R/bench.py --single-stream --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > $OUT/${TAG}_bench_under_rocprof.json 2> /dev/null
cp $OUT/kt/b_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
python3 $R/tools/prof_summary.py $OUT/kt/b_kernel_stats.csv 7 40 > $OUT/${TAG}_bench_kernel_stats_summary.txt
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $OUT/pmc$i -o p --output-format csv -- python3 $R/bench.py --single-stream --steps 2 --warmup 1 --no-cpu-baseline --no-pmc --no-variants > /dev/null 2>&1
  python3 $R/tools/pmc_summary.py $(ls $OUT/pmc$i/*counter_collection.csv | head -1) > $OUT/${TAG}_pmc_$i.csv
done
mv $OUT/${TAG}_pmc_1.csv $OUT/${TAG}_pmc_fetch_size.csv
mv $OUT/${TAG}_pmc_2.csv $OUT/${TAG}_pmc_write_size.csv
cat $OUT/${TAG}_pmc_3.csv > $OUT/${TAG}_pmc_mfma.csv; tail -n +2 $OUT/${TAG}_pmc_4.csv >> $OUT/${TAG}_pmc_mfma.csv; rm -f $OUT/${TAG}_pmc_3.csv $OUT/${TAG}_pmc_4.csv
rm -rf $OUT/kt $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4
ls -la $OUT
# secondary configurations: kernel-trace summaries only
for cfg in c3 c3r; do
  rocprofv3 --kernel-trace --stats -d $OUT/kt_$cfg -o b --output-format csv -- python3 $R/tools/config_bench.py $cfg > $OUT/${TAG}_${cfg}_log.txt 2>&1
  python3 $R/tools/config_bench.py --pmc $cfg 2>&1 | grep -E "pairs/s|HBM traffic" > $OUT/${TAG}_${cfg}_hbm.txt
  python3 $R/tools/prof_summary.py $OUT/kt_$cfg/b_kernel_stats.csv 11 30 > $OUT/${TAG}_${cfg}_kernel_stats_summary.txt
  tail -1 $OUT/${TAG}_${cfg}_log.txt | grep -v rocprofv3 >> $OUT/${TAG}_${cfg}_kernel_stats_summary.txt
  rm -rf $OUT/kt_$cfg $OUT/${TAG}_${cfg}_log.txt
done
rocprofv3 --kernel-trace --stats -d $OUT/kt_u -o b --output-format csv -- python3 $R/bench.py --single-stream --unpad --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --no-variants > $OUT/${TAG}_bench_unpad_under_rocprof.json 2> /dev/null
python3 $R/tools/prof_summary.py $OUT/kt_u/b_kernel_stats.csv 7 30 > $OUT/${TAG}_bench_unpad_kernel_stats_summary.txt
rm -rf $OUT/kt_u
ls -la $OUT
