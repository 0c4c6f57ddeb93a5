"""Concurrency summary of a rocprofv3 kernel trace of a multi-session run (development tool).

usage: overlap.py <p_kernel_trace.csv>
Prints, for the middle half of the trace: the time the device had 0, 1, 2, 3+ kernels in flight, the average number in flight,
and per kernel name the average duration and the average number of other kernels that were running beside it.
"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1], newline="") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|qtts::|void ", "", r["Kernel_Name"])[:70], r.get("Queue_Id", "")))
rows.sort()
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo, hi = t0 + (t1 - t0) // 4, t0 + 3 * (t1 - t0) // 4
sel = [r for r in rows if r[0] >= lo and r[1] <= hi]
ev = []
for s, e, _, _ in sel:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
level_time = defaultdict(int)
cur, last = 0, lo
area = 0
for t, d in ev:
    level_time[min(cur, 4)] += t - last
    area += cur * (t - last)
    cur += d
    last = t
span = hi - lo
print("window %.1f ms, %d kernels, queues %s" % (span / 1e6, len(sel), sorted(set(r[3] for r in sel))))
for k in sorted(level_time):
    print("  %d%s kernels in flight: %5.1f %%" % (k, "+" if k == 4 else "", 100.0 * level_time[k] / span))
print("  average in flight: %.2f" % (area / span))
# neighbours: sweep with active set size sampled at each kernel's midpoint (cheap approximation)
import bisect
starts = sorted(s for s, _, _, _ in sel)
ends = sorted(e for _, e, _, _ in sel)
per = defaultdict(lambda: [0, 0.0, 0.0])
for s, e, n, _ in sel:
    mid = (s + e) // 2
    active = bisect.bisect_right(starts, mid) - bisect.bisect_right(ends, mid)
    p = per[n]
    p[0] += 1
    p[1] += e - s
    p[2] += active - 1
print("%-72s %8s %8s %8s" % ("kernel", "calls", "avg_us", "beside"))
for n, (c, d, a) in sorted(per.items(), key=lambda kv: -kv[1][1])[:16]:
    print("%-72s %8d %8.2f %8.2f" % (n, c, d / c / 1e3, a / c))
