#!/bin/bash
R=$GRAFT_REPO_ROOT
cd $R/tools/abl
for amp in 0.1 0.3 1 3; do IA_ATTN_FWD=3 ./attn_dev.bin 256 577 12 0 0 $amp 0 0; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/kt13 -o t --output-format csv -- python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("/tmp/kt13/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
prev=None
out=[]
for i,r in enumerate(rows):
    if "attn_fwd3_kernel<false" in r["Kernel_Name"]:
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000
        p=rows[i-1]["Kernel_Name"][:40] if i else ""
        n=rows[i+1]["Kernel_Name"][:40] if i+1<len(rows) else ""
        out.append((d,p,n))
for d,p,n in out: print(f"{d:8.1f} us   prev: {p}   next: {n}")
PY
