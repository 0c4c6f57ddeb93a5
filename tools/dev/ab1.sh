#!/bin/bash
# A/B of environment switches with one stream alone (no codec, 200 frames): prints ms per frame per setting, two rounds.
set -o pipefail
for round in 1 2; do
for v in "X=0" "$@"; do
  env $v timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --pipeline 1 --frames 200 --no-codec --no-cpu-baseline > gpurun_out/ab1.log 2>&1 || { tail -5 gpurun_out/ab1.log; exit 1; }
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/ab1.log) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/ab1.log) $(grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/ab1.log)"
done
done
