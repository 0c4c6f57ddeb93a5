"""Developer scratch: per-frame kernel table from a rocprofv3 kernel_stats csv."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2])
print(f"{'kernel':72s} {'calls/fr':>8s} {'avg_us':>8s} {'us/frame':>9s}")
tot = 0
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    n = r['Name'].replace('void ', '').replace('qtts::', '').split('(')[0][:72]
    calls = int(r['Calls']); t = float(r['TotalDurationNs']) / 1e3
    tot += t
    print(f"{n:72s} {calls / frames:8.1f} {float(r['AverageNs']) / 1e3:8.2f} {t / frames:9.1f}")
print("sum us/frame", tot / frames)
