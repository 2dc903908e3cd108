#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=3
for v in 0 1 2 3; do
  export IA_ATTN_BWD=$v
  ./attn_dev.bin 2 255 4 1 0 1 0 | head -1
  ./attn_dev.bin 2 255 4 1 0 1 1 | head -1
  ./attn_dev.bin 2 577 3 1 0 1 0 | head -1
  ./attn_dev.bin 3 510 2 1 0 1 1 | head -1
  ./attn_dev.bin 2 129 2 1 0 1 1 | head -1
  ./attn_dev.bin 2 20 1 1 0 1 1 | head -1
  ./attn_dev.bin 2 64 1 1 0 1 0 | head -1
done
for rep in 1 2; do
for v in 0 3; do
  export IA_ATTN_BWD=$v
  ./attn_dev.bin 256 577 12 1 0 1 0 0
  ./attn_dev.bin 256 255 16 1 0 1 1 0
  ./attn_dev.bin 256 255 16 1 0.1 1 1 0
  ./attn_dev.bin 128 510 16 1 0 1 0 0
done
done
for v in 1 2; do
  export IA_ATTN_BWD=$v
  ./attn_dev.bin 256 577 12 1 0 1 0 0
  ./attn_dev.bin 256 255 16 1 0.1 1 1 0
done
