#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=3
for pad in 0 64000; do
  export IA_ATTN_PAD_LDS=$pad
  echo "===== pad=$pad"
  ./attn_dev_prof.bin 256 577 12 0 0 1 0 0
  ./attn_dev_prof.bin 256 255 16 0 0 1 0 0
done
