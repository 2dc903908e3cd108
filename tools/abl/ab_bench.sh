#!/bin/bash
# same-box A/B of the bench step: args = list of "NAME=VALUE" environment settings, one bench run (8 steps) each, twice round-robin
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for kv in "$@"; do
    v=$(env $kv python3 bench.py --steps 8 --warmup 3 --no-pmc --no-cpu-baseline --no-variants 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f pairs/s %.1f ms' % (d['value'], d['ms_per_step']))")
    echo "$kv: $v"
  done
done
