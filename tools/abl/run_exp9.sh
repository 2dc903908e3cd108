#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=3
for pad in 0 32000 64000; do
  export IA_ATTN_PAD_LDS=$pad
  for bin in attn_dev attn_dev_a4 attn_dev_a27; do
    echo "== $bin pad=$pad"
    ./$bin.bin 256 577 12 0 0 1 0 0
    ./$bin.bin 256 255 16 0 0 1 0 0
  done
done
