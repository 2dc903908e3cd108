#!/bin/bash
# final loss of a 3-step bench run per attention kernel combination: a NaN/inf here means a kernel poisons the weights
R=$GRAFT_REPO_ROOT
for combo in "2 0" "0 0" "2 3" "0 3" "2 1" "2 2"; do
set -- $combo
export IA_ATTN_FWD=$1 IA_ATTN_BWD=$2
for st in 1 2 4; do
echo -n "fwd=$1 bwd=$2 steps=$st: "
timeout 300 python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps $st --warmup 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d.get('final_loss'), d['ms_per_step'])"
done
done
