#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=2
for bv in 1 3; do
export IA_ATTN_BWD=$bv
for L in 385 577 193 608 255; do
echo "== bwd=$bv L=$L"
./attn_dev.bin 48 $L 12 1 0 1 0 40 | grep -v "rel err" | grep -c "scan: 0 bad"
./attn_dev.bin 48 $L 12 1 0 1 1 40 | grep -v "rel err" | grep -c "scan: 0 bad"
done; done
export IA_ATTN_BWD=3
for fv in 3 4; do
export IA_ATTN_FWD=$fv
for L in 385 577 193 255; do
echo "== fwd=$fv L=$L (forward output scan)"
./attn_dev.bin 48 $L 12 0 0 1 0 40 | grep -v "rel err" | grep -c "scan: 0 bad"
./attn_dev.bin 48 $L 12 0 0 1 1 40 | grep -v "rel err" | grep -c "scan: 0 bad"
done; done
