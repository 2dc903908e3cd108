#!/bin/bash
# wide shape sweep of the attention kernels in the torch-free harness: sampled fp32 reference + whole-output scans
cd "$(dirname "$0")"
worst=0
for fv in 3 4; do for L in 1 2 31 32 33 63 64 65 127 128 129 191 193 255 256 257 383 385 511 513 640 1000 1537 2047 2048; do for m in 0 1; do
  out=$(IA_ATTN_FWD=$fv IA_ATTN_BWD=3 ./attn_dev.bin 3 $L 2 1 0 1 $m 3 2>&1)
  echo "$out" | grep -q "scan: 0 bad" || { echo "FWD=$fv L=$L masked=$m: $out" | head -5; }
  echo "$out" | grep "rel err" | head -1
done; done; done | awk '{print} /rel err/{ for(i=1;i<=NF;i++) if($i=="err"||$i=="dq"||$i=="dk"||$i=="dv") { v=$(i+1)+0; if(v>w) w=v } } END{print "worst relative / absolute error over the sweep:", w}'
IA_ATTN_FWD=3 ./attn_dev.bin 2 2049 2 0 0 1 0 1 2>&1 | head -2
