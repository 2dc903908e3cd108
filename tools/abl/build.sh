#!/bin/bash
# builds attn_abl<n> (one binary per ablation level given on the command line) in this directory
cd "$(dirname "$0")"
for n in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DIA_ABL=$n $EXTRA -Wno-unused-value -I../../include attn_abl.hip -o attn_abl$n & done; wait
