#!/bin/bash
cd "$(dirname "$0")"
R=$GRAFT_REPO_ROOT
export IA_ATTN_FWD=3
echo "== random data"; ./attn_dev.bin 256 577 12 0 0 1 0 0
echo "== zero data";   ./attn_dev.bin 256 577 12 0 0 0 0 0
echo "== random, L=510"; ./attn_dev.bin 128 510 16 0 0 1 0 0
echo "== zero, L=510";   ./attn_dev.bin 128 510 16 0 0 0 0 0
cd /tmp && export TMPDIR=/tmp
for amp in 1 0; do
  rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $R/gpurun_out/clk_$amp -o x --output-format csv -- $R/tools/abl/attn_dev.bin 256 577 12 0 0 $amp 0 0 > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/clk_$amp/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "fwd3" in r["Kernel_Name"] and r["Counter_Name"]=="GRBM_GUI_ACTIVE"]
k=glob.glob("$R/gpurun_out/clk_$amp/**/*kernel_trace.csv",recursive=True)[0]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(k)) if "fwd3" in r["Kernel_Name"]]
import statistics
g=statistics.mean(float(r["Counter_Value"]) for r in rows); t=statistics.mean(d)
print("amp=$amp GRBM_GUI_ACTIVE mean", g, "duration ns", t, "-> GHz (if per-XCD sum of 8: /8)", g/t, g/t/8)
PY
done
