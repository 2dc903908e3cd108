#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/item_alignment_amd/libitemalign_hip.so /tmp/lib_default.so
for v in default k0; do
  if [ $v != default ]; then cp $R/tools/abl/lib_$v.so $R/item_alignment_amd/libitemalign_hip.so; fi
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/wf_$v -o p --output-format csv -- python3 $R/tools/abl/wgrad_fetch.py > /dev/null 2>&1
  echo "== $v"; python3 $R/tools/pmc_summary.py $(ls $R/gpurun_out/wf_$v/*counter_collection.csv | head -1) | grep -E "t256w|kernel,"
  cp /tmp/lib_default.so $R/item_alignment_amd/libitemalign_hip.so
done
