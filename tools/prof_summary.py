"""Condense a rocprofv3 --kernel-trace --stats kernel_stats CSV: per-step calls, average duration, share."""
import csv, sys
path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/steps/1e6:.2f} ms/step over {steps} steps")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:78]
    print("%6.2f%% %6.1f/step avg %8.1f us  %s" % (float(r["Percentage"]), int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, n))
