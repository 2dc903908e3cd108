#!/bin/bash
# builds attn_dev.bin and the ablation binaries attn_dev_a<N>.bin (-DIA_F3_ABL=N) given as arguments; "p0" = IA_F3_PRESCALE=0
cd "$(dirname "$0")"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -I../../include -I../../item_alignment_amd/csrc"
/opt/rocm/bin/hipcc $FLAGS attn_dev.hip -o attn_dev.bin &
for n in "$@"; do
  if [ "$n" = "r0" ]; then /opt/rocm/bin/hipcc $FLAGS -DIA_F3_ROT=0 attn_dev.hip -o attn_dev_r0.bin &
  elif [ "$n" = "r0a4" ]; then /opt/rocm/bin/hipcc $FLAGS -DIA_F3_ROT=0 -DIA_F3_ABL=4 attn_dev.hip -o attn_dev_r0a4.bin &
  elif [ "$n" = "p0" ]; then /opt/rocm/bin/hipcc $FLAGS -DIA_F3_PRESCALE=0 attn_dev.hip -o attn_dev_p0.bin &
  else /opt/rocm/bin/hipcc $FLAGS -DIA_F3_ABL=$n attn_dev.hip -o attn_dev_a$n.bin & fi
done
wait
