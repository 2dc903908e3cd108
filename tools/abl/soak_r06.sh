#!/bin/bash
# round-6 soak: long runs on the round's new code paths; every final loss must be finite.
#   bench, 300 steps (fused attention backward with the new exchange layout, claim-ordered GEMMs off / on)
#   bench, 100 steps with IA_ATTN_EXACT_DELTA=1 (delta pre-pass + the dQ / dK,dV pair in every layer)
#   C3, 200 steps (direct convolutions with lane-offset DMA, ECA pooling through conv3, fused tail activation), C3 on resnetv2_50
#   C5x, 40 steps at 64 pairs (SwiGLU recompute, transposed shadows for the multimodal Linears)
#   resnetv2_50x1_bitm 200 steps, resnetv2_50x3_bitm_in21k 40 steps (GroupNorm kernels, standardised weights, zero-ring stem)
R=${GRAFT_REPO_ROOT:-.}
cd $R
for d in 0 1; do
IA_GEMM_DYNAMIC=$d python3 bench.py --steps 300 --warmup 10 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys,json,math; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 300 steps, IA_GEMM_DYNAMIC=$d:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step, final loss', d['final_loss'], 'finite' if math.isfinite(d['final_loss']) else 'NOT FINITE')"
done
IA_ATTN_EXACT_DELTA=1 python3 bench.py --steps 100 --warmup 5 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys,json,math; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 100 steps, IA_ATTN_EXACT_DELTA=1:', round(d['value'],1), 'pairs/s', round(d['ms_per_step'],1), 'ms/step, final loss', d['final_loss'], 'finite' if math.isfinite(d['final_loss']) else 'NOT FINITE')"
IA_CB_STEPS=200 python3 tools/config_bench.py c3 2>&1 | grep -E "pairs/s"
IA_CB_STEPS=200 python3 tools/config_bench.py c3r 2>&1 | grep -E "pairs/s"
IA_CB_STEPS=40 python3 tools/config_bench.py c5x 2>&1 | grep -E "pairs/s"
# (added with the BiT towers and the strided-convolution kernels: C3 above runs through them; the BiT towers below)
IA_CB_STEPS=200 python3 tools/config_bench.py c3b 2>&1 | grep -E "pairs/s"
IA_CB_STEPS=40 python3 tools/config_bench.py c3b3 2>&1 | grep -E "pairs/s"
