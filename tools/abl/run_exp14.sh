#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for fv in 0 2; do
export IA_ATTN_FWD=$fv
rm -rf /tmp/kt14
rocprofv3 --kernel-trace -d /tmp/kt14 -o t --output-format csv -- python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 3 --warmup 1 > /dev/null 2>&1
echo "=== IA_ATTN_FWD=$fv"
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/kt14/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
# split into steps by adamw kernel
steps=[[]]
for r in rows:
    steps[-1].append(r)
    if "adamw_kernel" in r["Kernel_Name"]: steps.append([])
for si,st in enumerate(steps[:4]):
    agg=collections.defaultdict(lambda:[0,0.0])
    for r in st:
        n=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:48]
        d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000
        agg[n][0]+=1; agg[n][1]+=d
    tot=sum(v[1] for v in agg.values())
    print(f"-- step {si}: total kernel time {tot/1000:.1f} ms")
    for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:14]:
        print(f"   {n:50s} {c:4d} x {t/c:8.1f} us")
PY
done
