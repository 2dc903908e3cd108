#!/bin/bash
# timing ablations of the fused attention backward at the bench shape (the outputs of the ablated builds are wrong on purpose)
cd "$(dirname "$0")"
for n in 0 1 2 4 8 16 32 64 128 255; do
  echo "abl=$n: $(timeout 120 ./attn_abl_f$n.bin 512 255 16 1 0 1 1 0 2>&1 | tail -1)"
done
echo "abl=0 dropout: $(timeout 120 ./attn_abl_f0.bin 512 255 16 1 0.1 1 1 0 2>&1 | tail -1)"
echo "== PMC (fused kernel only)"
$GRAFT_REPO_ROOT/tools/pmc_attn.sh $GRAFT_REPO_ROOT/tools/abl/attn_abl_f0.bin 512 255 16 1 0 1 1 0 2>&1 | grep -i "fused\|kernel" | head -40
