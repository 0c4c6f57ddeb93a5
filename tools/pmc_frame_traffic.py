#!/usr/bin/env python3
"""HBM-side traffic of one decode frame from rocprofv3 PMC passes (tools/dev/pmc_frame.sh): per-kernel sums of FETCH_SIZE and
WRITE_SIZE (KB) of an eager single-stream bench run at two frame counts; the difference divided by the extra frames is the
traffic of one frame (prefill, weight initialisation and the profiling pass after the timed region cancel).

gfx950 correction: FETCH_SIZE counts a 128-byte request of a 16-byte-per-lane read as 64 bytes, so the read side is doubled;
WRITE_SIZE is exact for 16-byte-per-lane stores.

    python tools/pmc_frame_traffic.py gpurun_out 12 22 alone > /tmp/alone.json      (then the same for `shared`; the two
    objects go into bench_data/frame_traffic.json as {"alone": .., "shared": ..}, which bench.py quotes by kernel selection)"""
import json
import sys


def load(path):
    rows = {}
    for line in open(path):
        v, n, k = line.rstrip("\n").split("\t")
        rows[k] = (float(v), int(n))
    return rows


def main():
    d, f0, f1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    tag = sys.argv[4] if len(sys.argv) > 4 else "alone"
    frames = f1 - f0
    out = {"frames_low": f0, "frames_high": f1, "unit": "bytes per decode frame (single stream, eager launches)", "kernel_selection": tag,
           "command": "rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 bench.py --no-graph --steps 1 --warmup 0 "
                      "--pipeline 1 --no-codec --no-cpu-baseline --frames <N>",
           "correction": "read side = 2 x FETCH_SIZE (gfx950), write side = WRITE_SIZE"}
    per_kernel = {}
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        a, b = load(f"{d}/pmc_frame_{tag}_{c}_{f0}.sum"), load(f"{d}/pmc_frame_{tag}_{c}_{f1}.sum")
        scale = 2.0 if c == "FETCH_SIZE" else 1.0
        tot[c] = 0.0
        for k in b:
            dv = (b[k][0] - a.get(k, (0.0, 0))[0]) * 1024.0 * scale / frames
            dn = (b[k][1] - a.get(k, (0.0, 0))[1]) / frames
            if dn <= 0:
                continue
            per_kernel.setdefault(k, {"launches_per_frame": dn})["read_bytes" if c == "FETCH_SIZE" else "write_bytes"] = round(dv)
            tot[c] += dv
    out["read_bytes"] = round(tot["FETCH_SIZE"])
    out["write_bytes"] = round(tot["WRITE_SIZE"])
    out["traffic_bytes"] = out["read_bytes"] + out["write_bytes"]
    out["launches_per_frame"] = round(sum(v["launches_per_frame"] for v in per_kernel.values()), 1)
    out["kernels"] = dict(sorted(per_kernel.items(), key=lambda kv: -(kv[1].get("read_bytes", 0) + kv[1].get("write_bytes", 0)))[:16])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
