"""Developer scratch: sum a rocprofv3 --pmc counter over all dispatches of a counter_collection csv."""
import csv, sys
tot = 0.0
n = 0
by = {}
for r in csv.DictReader(open(sys.argv[1])):
    v = float(r["Counter_Value"])
    tot += v
    n += 1
    k = r["Kernel_Name"].replace("void ", "").replace("qtts::", "").split("(")[0][:60]
    by[k] = by.get(k, 0.0) + v
print("dispatch-rows", n, "total", tot)
for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  {k:60s} {v:16.1f}")
