#!/bin/bash
cd "$(dirname "$0")"
for bv in 1 0 2; do
export IA_ATTN_BWD=$bv IA_ATTN_FWD=2
for rep in 1 2 3; do ./attn_dev.bin 256 577 12 1 0 1 0 2 | grep -v "rel err"; done
./attn_dev.bin 256 255 16 1 0 1 1 2 | grep -v "rel err"
./attn_dev.bin 64 577 12 1 0 1 0 2 | grep -v "rel err"
done
