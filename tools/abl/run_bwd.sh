#!/bin/bash
# fwd + bwd timing of attn_abl<n> on the bench shapes (args: ablation levels)
cd "$(dirname "$0")"
for n in "$@"; do
  for shape in "256 255 16" "256 577 12"; do
    ./attn_abl$n $shape 0 | sed "s/abl=0/abl=$n/"; ./attn_abl$n $shape 1 | sed "s/abl=0/abl=$n/"
    ./attn_abl$n $shape 0 0.1 | sed "s/abl=0/abl=$n drop/"; ./attn_abl$n $shape 1 0.1 | sed "s/abl=0/abl=$n drop/"
  done
done
