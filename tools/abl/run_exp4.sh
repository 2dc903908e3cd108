#!/bin/bash
cd "$(dirname "$0")"
for rep in 1 2; do
for v in 3 2; do
  export IA_ATTN_FWD=$v
  ./attn_dev.bin 256 255 16 0 0.1 1 1 0
  ./attn_dev.bin 256 255 16 0 0.1 1 0 0
  ./attn_dev.bin 256 255 16 0 0 1 1 0
done
done
