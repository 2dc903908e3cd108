#!/bin/bash
# forward ablations of the round-3 kernel at the bench shapes.  args: IA_ATTN_FWD versions
cd "$(dirname "$0")"
for v in "$@"; do
  export IA_ATTN_FWD=$v
  ./attn_dev.bin 2 255 4 0 0 1 1 | head -1
  ./attn_dev.bin 2 577 3 0 0 1 0 | head -1
  ./attn_dev.bin 3 510 2 0 0 1 1 | head -1
  ./attn_dev.bin 2 200 2 0 0 4 1 | head -1
  ./attn_dev_p0.bin 2 577 3 0 0 1 0 | head -1
  ./attn_dev_p0.bin 2 200 2 0 0 4 1 | head -1
  ./attn_dev_p0.bin 2 200 2 0 0 12 1 | head -1
  for bin in attn_dev attn_dev_p0 attn_dev_a1 attn_dev_a2 attn_dev_a3 attn_dev_a4 attn_dev_a8 attn_dev_a16 attn_dev_a24 attn_dev_a27; do
    echo "== $bin"
    ./$bin.bin 256 577 12 0 0 1 0 0
    ./$bin.bin 256 255 16 0 0 1 0 0
  done
done
