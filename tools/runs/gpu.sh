#!/bin/bash
# gpu.sh <timeout-seconds> <command...>: one gpurun call, retried while the pod has no free slot (exit code 3 / "transient")
t=$1; shift
for i in 1 2 3 4 5 6 7 8 9 10; do
  out=$(/usr/local/graft/bin/gpurun --timeout "$t" -- "$@" 2>&1)
  echo "$out" | tail -40
  if echo "$out" | grep -q "status=transient"; then sleep 90; continue; fi
  break
done
