#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
python3 bench.py --steps 1500 --warmup 10 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "
import sys,json,math; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench 1500 steps:', d['value'], 'pairs/s', d['ms_per_step'], 'ms/step, final loss', d['final_loss'], 'finite' if math.isfinite(d['final_loss']) else 'NOT FINITE')"
cd tools/abl
for args in "256 577 12 1 0 1 0" "512 255 16 1 0.1 1 1" "256 577 12 0 0 1 0" "512 255 16 0 0.1 1 1" "128 385 12 1 0 1 0" "128 193 12 1 0 1 1"; do
  out=$(./attn_dev.bin $args 400 2>&1)
  echo "attn_dev $args x400 scans: $(echo "$out" | grep -c 'scan: 0 bad') clean, $(echo "$out" | grep 'scan:' | grep -vc 'scan: 0 bad') bad"
done
