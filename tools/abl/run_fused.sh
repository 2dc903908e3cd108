#!/bin/bash
# round 4: the fused one-kernel attention backward (IA_ATTN_BWD=7, default) against the round-3 kernel pair (IA_ATTN_BWD=3)
cd "$(dirname "$0")"
echo "== correctness (sampled (sequence, head) pairs vs fp32 host reference; whole-output scan)"
for args in "2 33 1 1 0 1 0 2" "2 64 2 1 0 1 1 2" "3 65 1 1 0 1 0 2" "1 129 3 1 0 1 1 2" "2 220 16 1 0 1 1 2" "4 255 16 1 0 1 1 2" "4 255 16 1 0 1 0 2" "3 256 4 1 0 2 1 2" "40 255 16 1 0 1 1 3"; do
  timeout 120 ./attn_dev.bin $args 2>&1 | grep -v "^$" | head -8
done
echo "== timing at the bench shape (512 x 255 x 16), masked ragged batch and full"
for v in 7 3; do
  for args in "512 255 16 1 0 1 1 0" "512 255 16 1 0.1 1 1 0" "512 255 16 1 0 1 0 0" "512 255 16 1 0.1 1 0 0" "512 220 16 1 0 1 1 0"; do
    IA_ATTN_BWD=$v timeout 120 ./attn_dev.bin $args 2>&1 | tail -1
  done
done
