#!/bin/bash
# Single-session frame time and three-session throughput (no codec) under a list of environment variants ("-" = none).
set -o pipefail
out=gpurun_out
i=0
for v in "$@"; do
    [ "$v" = "-" ] && v="QTTS_NOP=1"
    env $v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-codec --no-cpu-baseline --frames 200 > $out/bm1_$i.json 2>/dev/null || exit 2
    env $v timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-codec --no-cpu-baseline --frames 200 > $out/bm3_$i.json 2>/dev/null || exit 3
    python - "$v" $out/bm1_$i.json $out/bm3_$i.json <<'PY'
import json, sys
a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
print("%-40s one session %.3f ms/frame   three sessions %.1f audio-s/s (%.3f ms/frame)" % (sys.argv[1], a["ms_per_frame"], b["value"], b["ms_per_frame"]), flush=True)
PY
    i=$((i+1))
done
