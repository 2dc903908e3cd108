"""Per-kernel sums of a rocprofv3 --pmc counter_collection CSV (one row per kernel name and counter)."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
print("kernel,counter,launches,sum,mean_per_launch")
for k, v in sorted(agg.items(), key=lambda kv: -max(kv[1].values())):
    for c, s in v.items():
        n = cnt[(k, c)]
        print(f"\"{k}\",{c},{n},{s:.0f},{s/n:.1f}")
