#!/bin/bash
R=$GRAFT_REPO_ROOT
python3 $R/tools/abl/gemm_ksweep.py 65280 1024
python3 $R/tools/abl/gemm_ksweep.py 16320 4096
IA_GEMM_DBG=64 python3 $R/tools/abl/gemm_ksweep.py 65280 1024
for s in 0 1 0 1; do
echo -n "IA_TOWER_STREAMS=$s: "
IA_TOWER_STREAMS=$s python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['final_loss'])"
done
