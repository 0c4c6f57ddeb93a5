#!/bin/bash
# rows per launch x sessions in flight, no codec, 200 frames: audio-s/s and ms per frame
set -o pipefail
for cfg in "$@"; do
  set -- $cfg
  timeout -k 10 300 python -u bench.py --batch $1 --pipeline $2 --steps $((2 * $2)) --warmup $2 --frames 200 --no-codec --no-cpu-baseline > gpurun_out/bm.log 2>gpurun_out/bm.err || { tail -3 gpurun_out/bm.err; exit 1; }
  echo "batch $1 x $2 sessions: $(grep -o '"value": [0-9.]*' gpurun_out/bm.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/bm.log)"
done
