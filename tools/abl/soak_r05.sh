#!/bin/bash
# round-5 soak: long bench runs with the static and the claim-ordered persistent GEMMs (both towers in flight), a long C3 run on the
# direct convolution kernels; every final loss must be finite, the two bench runs must agree to the noise of the atomically
# accumulated embedding-table gradients
R=${GRAFT_REPO_ROOT:-.}
cd $R
for d in 0 1; do
IA_GEMM_DYNAMIC=$d python3 bench.py --steps 300 --warmup 10 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys,json,math; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 300 steps, IA_GEMM_DYNAMIC=$d:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step, final loss', d['final_loss'], 'finite' if math.isfinite(d['final_loss']) else 'NOT FINITE')"
done
IA_CB_STEPS=200 python3 tools/config_bench.py c3 2>&1 | grep -E "pairs/s"
IA_CB_STEPS=200 python3 tools/config_bench.py c3r 2>&1 | grep -E "pairs/s"
