#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/runs/run.sh suite
O=gpurun_out/r06_ab_fused_exchange.txt; : > $O
echo "attn_bwd_fused_kernel<true> at 512 x 255 x 16, dropout 0.1 (tools/attn_one.py under rocprofv3): this commit's exchange layout (dense rows + XOR key)" >> $O
echo "against the previous one (72-byte rows), two libraries on one box, interleaved" >> $O
cp item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
for rep in 1 2; do
  for which in default prev; do
    if [ $which = prev ]; then cp tools/abl/lib_prev_exchange.so item_alignment_amd/libitemalign_hip.so; else cp /tmp/lib_default.so item_alignment_amd/libitemalign_hip.so; fi
    TAG=r06tmp bash tools/runs/run.sh prof bwdf_$which python3 tools/attn_one.py 512 255 16 0.1 > /dev/null 2>&1
    echo "$which (rep $rep): $(grep -E 'attn_bwd_fused' gpurun_out/r06tmp_bwdf_${which}_kernel_stats_summary.txt)" >> $O
  done
done
cp /tmp/lib_default.so item_alignment_amd/libitemalign_hip.so
cat $O
bash tools/runs/run.sh quick
