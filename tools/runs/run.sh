#!/bin/bash
# One parameterised entry for every command list that runs ON the GPU box (through tools/runs/gpu.sh -> gpurun).  It replaces the
# 21 one-off r05_run<N>.sh scripts of round 5 (they are in the git history).
#   tools/runs/run.sh suite                      the whole -m gpu suite (tail of the log -> gpurun_out/<tag>_gpu_suite.txt)
#   tools/runs/run.sh tests <pytest args...>     part of it
#   tools/runs/run.sh bench [bench.py args...]   one bench line -> gpurun_out/<tag>_bench_line.json, key figures printed
#   tools/runs/run.sh quick                      short bench (no variants / PMC / CPU baseline), value + frac printed
#   tools/runs/run.sh prof <name> <cmd...>       rocprofv3 --kernel-trace --stats of <cmd> -> gpurun_out/<tag>_<name>_kernel_stats_summary.txt
#   tools/runs/run.sh profiles                   tools/collect_profiles.sh <tag> (bench kernel stats + PMC passes + c3 / c3r + unpad)
#   tools/runs/run.sh configs [c2 c3 c4 c5x...]  tools/config_bench.py lines -> gpurun_out/<tag>_config_bench.txt
#   tools/runs/run.sh ab <alt.so> <cmd...>       same-box A/B of two library builds (tools/abl/lib_ab.sh)
#   tools/runs/run.sh seq <file>                 run every line of <file> as one of the above, in order (several steps in one gpurun call)
# TAG (environment, default r06) prefixes the outputs.
TAG=${TAG:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" && mkdir -p gpurun_out
sub=$1; shift
case "$sub" in
  suite)
    python -m pytest tests -m gpu -x -q --tb=short "$@" > gpurun_out/${TAG}_gpu_suite_full.txt 2>&1      # (the whole log: gpurun only echoes a tail)
    grep -E "^(FAILED|ERROR)|passed|failed|^E  " gpurun_out/${TAG}_gpu_suite_full.txt | cut -c1-600 | tail -25 > gpurun_out/${TAG}_gpu_suite.txt; cat gpurun_out/${TAG}_gpu_suite.txt ;;
  tests)
    python -m pytest "$@" -q -x 2>&1 | grep -E "passed|failed|error|^E  |^FAILED" | cut -c1-300 ;;
  bench)
    python bench.py "$@" > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench_line.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_line.json"))
r = d.get("roofline") or {}
print(d["value"], d["ms_per_step"], "frac", r.get("frac"), "traffic", r.get("traffic"), "whole", d.get("mfma_fraction_whole_step"), "dense", d.get("mfma_fraction_dense"))
print({k: round(v["value"], 1) for k, v in (d.get("variants") or {}).items()})
print(d.get("cpu_baseline"))
PY
    ;;
  quick)
    python bench.py --steps ${STEPS:-8} --warmup 2 --no-cpu-baseline --no-pmc --no-variants "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('pairs/s', round(d['value'],1), 'ms', round(d['ms_per_step'],2), 'frac', round((d.get('roofline') or {}).get('frac') or 0, 4))" ;;
  prof)
    name=$1; shift
    export TMPDIR=/tmp; rm -rf /tmp/kt_$name
    args=(); for a in "$@"; do if [ -f "$R/$a" ]; then args+=("$R/$a"); else args+=("$a"); fi; done      # rocprofv3 runs from /tmp: repo files by absolute path
    (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o b --output-format csv -- "${args[@]}" > /tmp/kt_$name.log 2>&1)
    python3 tools/prof_summary.py $(find /tmp/kt_$name -name "*kernel_stats.csv" | head -1) ${PROF_STEPS:-1} 40 > gpurun_out/${TAG}_${name}_kernel_stats_summary.txt
    tail -3 /tmp/kt_$name.log | grep -v rocprofv3 >> gpurun_out/${TAG}_${name}_kernel_stats_summary.txt
    cat gpurun_out/${TAG}_${name}_kernel_stats_summary.txt ;;
  profiles)
    bash tools/collect_profiles.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1; ls gpurun_out/profiles | grep $TAG ;;
  configs)
    python tools/config_bench.py "$@" 2>&1 | grep -E "pairs/s" | tee gpurun_out/${TAG}_config_bench.txt ;;
  ab)
    alt=$1; shift
    bash tools/abl/lib_ab.sh "$alt" "$@" ;;
  seq)
    while IFS= read -r line; do
      [ -z "$line" ] && continue
      case "$line" in \#*) continue ;; esac
      echo "=== $line"
      bash "$0" $line
    done < "$1" ;;
  *)
    echo "usage: see the header of $0" >&2; exit 2 ;;
esac
