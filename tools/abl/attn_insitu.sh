#!/bin/bash
# in-situ attention kernel times (bench, steady-state steps) per kernel generation
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for combo in "0 3" "2 0" "3 3" "4 3"; do
set -- $combo
export IA_ATTN_FWD=$1 IA_ATTN_BWD=$2
rm -rf /tmp/kt24
rocprofv3 --kernel-trace -d /tmp/kt24 -o t --output-format csv -- python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 4 --warmup 1 > /tmp/b24.json 2>/dev/null
echo "=== IA_ATTN_FWD=$1 IA_ATTN_BWD=$2  $(python3 -c "import json;d=json.loads(open('/tmp/b24.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d.get('final_loss'))")"
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/kt24/**/*kernel_trace.csv",recursive=True)[0]
rows=sorted(csv.DictReader(open(f)), key=lambda r:int(r["Start_Timestamp"]))
steps=[[]]
for r in rows:
    steps[-1].append(r)
    if "adamw_kernel" in r["Kernel_Name"]: steps.append([])
agg=collections.defaultdict(lambda:[0,0.0])
for st in steps[2:5]:
    for r in st:
        n=r["Kernel_Name"].replace("void (anonymous namespace)::","")[:40]
        if "attn" not in n: continue
        agg[n][0]+=1; agg[n][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1000
tot=0
for n,(c,t) in sorted(agg.items()):
    print(f"   {n:42s} {c//3:3d}/step x {t/c:8.1f} us"); tot+=t/3
print(f"   attention total {tot/1000:.2f} ms/step")
PY
done
