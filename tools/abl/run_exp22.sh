#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_BWD=1 IA_ATTN_FWD=2
for d in "$@"; do
for L in 385 577 449; do
echo "== DBG=$d L=$L"
./attn_dev_d$d.bin 48 $L 12 1 0 1 0 40 | grep -v "rel err" | grep -c "scan: 0 bad"
./attn_dev_d$d.bin 48 $L 12 1 0 1 0 40 | grep -v "rel err" | grep "bad" | grep -v "scan: 0" | head -4
done; done
