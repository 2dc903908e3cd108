#!/bin/bash
# fwd-only timing of attn_abl<n> on the bench shapes, three interleaved repeats (args: ablation levels)
cd "$(dirname "$0")"
for rep in 1 2 3; do
for n in "$@"; do
  for shape in "256 255 16" "256 577 12"; do
    ./attn_abl$n $shape 0 | sed "s/abl=0/abl=$n/"
  done
done
done
