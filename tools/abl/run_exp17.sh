#!/bin/bash
R=$GRAFT_REPO_ROOT
for bv in 0 1; do for args in "128" "4" "128 nodrop"; do
IA_ATTN_BWD=$bv timeout 300 python3 $R/tools/abl/dbg_grads.py $args 2>&1 | tail -30
done; done
