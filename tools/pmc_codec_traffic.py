#!/usr/bin/env python3
"""HBM-side traffic of one codec decode from the per-kernel counter sums tools/dev/pmc_codec.sh leaves (FETCH_SIZE / WRITE_SIZE in KB).

    python tools/pmc_codec_traffic.py gpurun_out 16 375 > bench_data/codec_traffic.json"""
import json
import sys

out, batch, frames = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
SKIP = ("pack_weight", "at::native", "cast_rows", "fillBuffer", "copyBuffer", "rocclr")


def sums(counter):
    d = {}
    for line in open(f"{out}/pmc_codec_{counter}.sum"):
        v, n, k = line.rstrip("\n").split("\t")
        if not any(s in k for s in SKIP):
            d[k] = (float(v) * 1024.0, int(n))
    return d


rd, wr = sums("FETCH_SIZE"), sums("WRITE_SIZE")
kernels = {}
for k in sorted(set(rd) | set(wr), key=lambda k: -(2 * rd.get(k, (0, 0))[0] + wr.get(k, (0, 0))[0])):
    kernels[k] = dict(launches=max(rd.get(k, (0, 0))[1], wr.get(k, (0, 0))[1]), read_bytes=int(2 * rd.get(k, (0, 0))[0]),
                      write_bytes=int(wr.get(k, (0, 0))[0]))
read = sum(v["read_bytes"] for v in kernels.values())
write = sum(v["write_bytes"] for v in kernels.values())
print(json.dumps({
    "unit": f"bytes per codec decode of {batch} utterances x {frames} frames (bf16, reference-default shapes, one group)",
    "command": f"B={batch} REPS=1 rocprofv3 --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python3 tools/dev/dev_codec_prof.py  (tools/dev/pmc_codec.sh)",
    "correction": "read side = 2 x FETCH_SIZE (gfx950), write side = WRITE_SIZE; weight upload / initialisation kernels left out",
    "batch": batch, "frames": frames, "read_bytes": read, "write_bytes": write, "traffic_bytes": read + write, "kernels": kernels}, indent=1))
