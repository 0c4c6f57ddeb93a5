#!/usr/bin/env python3
"""Per-frame kernel table from a rocprofv3 `--kernel-trace --stats --output-format csv` run of bench.py.

    python tools/profile_summary.py <kernel_stats.csv> <decode frames in the run> "<command profiled>" > profiles/rNN_....txt"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2])
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
top = int(sys.argv[4]) if len(sys.argv) > 4 else 30
print(f"# rocprofv3 --kernel-trace --stats of `{cmd}`")
print(f"# per-frame columns divide by the {int(frames)} decode frames of the run")
print(f"{'kernel':78s} {'calls':>8s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s} {'calls/frame':>11s} {'us/frame':>9s}")
tot_calls = tot_us = 0.0
for i, r in enumerate(rows):
    n = r['Name'].replace('void ', '').replace('qtts::', '').replace('(anonymous namespace)::', '')
    n = re.split(r'\(', n)[0][:78]
    calls = int(r['Calls'])
    us = float(r['TotalDurationNs']) / 1e3
    tot_calls += calls
    tot_us += us
    if i < top:
        print(f"{n:78s} {calls:8d} {us / 1e3:9.2f} {float(r['AverageNs']) / 1e3:8.2f} {float(r['Percentage']):6.2f} {calls / frames:11.1f} {us / frames:9.1f}")
print(f"# all kernels: {tot_calls / frames:.1f} launches and {tot_us / frames:.1f} us of kernel time per frame")
