#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_BWD=1 IA_ATTN_FWD=2
for L in 705 833 769 800 641 385 193; do
echo "== L=$L nkt=$(( (L+63)/64 )) last slot $(( ((L+63)/64 - 1) % 3 ))"
./attn_dev_d0.bin 48 $L 12 1 0 1 0 40 | grep -v "rel err" | grep -c "scan: 0 bad"
./attn_dev_d0.bin 48 $L 12 1 0 1 0 40 | grep -v "rel err" | grep "bad" | grep -v "scan: 0" | head -4
done
