#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R
for combo in "2 0" "0 3" "0 3" "2 3" "0 0" "0 3"; do set -- $combo
echo -n "IA_ATTN_FWD=$1 IA_ATTN_BWD=$2: "; IA_ATTN_FWD=$1 IA_ATTN_BWD=$2 python3 tools/config_bench.py c5x 2>&1 | grep "pairs/s" | sed 's/.*step, //'
done
