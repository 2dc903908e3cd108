#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=3
./attn_dev.bin 2 255 4 0 0 1 0 | head -1
./attn_dev.bin 2 255 4 0 0 1 1 | head -1
./attn_dev.bin 2 577 3 0 0 1 0 | head -1
./attn_dev.bin 3 510 2 0 0 1 1 | head -1
./attn_dev.bin 2 129 2 0 0 1 1 | head -1
./attn_dev.bin 2 20 1 0 0 1 1 | head -1
./attn_dev.bin 2 64 1 0 0 1 0 | head -1
./attn_dev.bin 2 200 2 0 0 4 1 | head -1
for rep in 1 2; do
for bin in attn_dev attn_dev_prev; do
  echo "== $bin"
  ./$bin.bin 256 577 12 0 0 1 0 0
  ./$bin.bin 256 255 16 0 0 1 0 0
  ./$bin.bin 256 255 16 0 0.1 1 1 0
  ./$bin.bin 128 510 16 0 0 1 0 0
done
done
