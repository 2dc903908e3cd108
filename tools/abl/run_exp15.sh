#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pm15
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d /tmp/pm15 -o t --output-format csv -- python3 $R/bench.py --no-pmc --no-cpu-baseline --no-variants --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pm15/**/*counter_collection.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "attn_fwd3_kernel<false" in r["Kernel_Name"]]
by=collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"],{})[r["Counter_Name"]]=float(r["Counter_Value"])
for d,c in by.items():
    print(d, " ".join(f"{k}={v/1e6:.1f}M" for k,v in sorted(c.items())))
PY
