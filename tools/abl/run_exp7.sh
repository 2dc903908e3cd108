#!/bin/bash
cd "$(dirname "$0")"
export IA_ATTN_FWD=3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 3 0; do
  export IA_ATTN_BWD=$v
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/st_bwd$v -o x --output-format csv -- $R/tools/abl/attn_dev.bin 256 577 12 1 0 1 0 0 > /dev/null 2>&1
  echo "== stats bwd=$v ViT"; python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/st_bwd$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/st_bwdt$v -o x --output-format csv -- $R/tools/abl/attn_dev.bin 256 255 16 1 0.1 1 1 0 > /dev/null 2>&1
  echo "== stats bwd=$v text dropout"; python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/st_bwdt$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
done
export IA_ATTN_BWD=3
echo "== PMC new bwd, ViT shape"
$R/tools/pmc_attn.sh $R/tools/abl/attn_dev.bin 256 577 12 1 0 1 0 0 | grep -v fillBuffer
