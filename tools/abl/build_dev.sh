#!/bin/bash
# builds the attention harness attn_dev.bin (and, with the argument "p0", attn_dev_p0.bin: IA_F3_PRESCALE=0, the exact-exponent variant)
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../include -I../../item_alignment_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS attn_dev.hip -o attn_dev.bin &
for n in "$@"; do
  if [ "$n" = "p0" ]; then /opt/rocm/bin/hipcc $FLAGS -DIA_F3_PRESCALE=0 attn_dev.hip -o attn_dev_p0.bin & fi
done
wait
