#!/bin/bash
# pairs per GPU vs throughput (same step; memory grows ~0.8 GB per pair)
R=$GRAFT_REPO_ROOT
for p in 128 192 256; do
echo -n "pairs_per_gpu=$p: "
timeout 600 python3 $R/bench.py --pairs-per-gpu $p --resident-batches 6 --no-pmc --no-cpu-baseline --no-variants --steps 6 --warmup 2 2>/tmp/err_$p.txt | python3 -c "
import sys,json
t=sys.stdin.read().strip().splitlines()
if not t: print('failed:', open('/tmp/err_$p.txt').read()[-300:])
else:
    d=json.loads(t[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'])"
python3 -c "import torch" 2>/dev/null
done
