#!/bin/bash
cd "$(dirname "$0")"
for bin in attn_dev.bin attn_dev_swap.bin; do
echo "=== $bin"
for args in "2 33 1 1 0 1 0 2" "3 65 1 1 0 1 0 2" "1 129 3 1 0 1 1 2" "2 220 16 1 0 1 1 2" "4 255 16 1 0 1 1 2" "40 255 16 1 0 1 1 3" "600 255 16 1 0 1 0 3"; do
  timeout 120 ./$bin $args 2>&1 | grep -v "^$" | grep -v "scan: 0 bad" | head -8
done
for args in "512 255 16 1 0 1 1 0" "512 255 16 1 0.1 1 1 0" "512 255 16 1 0 1 0 0" "512 255 16 1 0.1 1 0 0" "512 220 16 1 0 1 1 0"; do
  timeout 120 ./$bin $args 2>&1 | tail -1
done
done
