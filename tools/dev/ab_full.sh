#!/bin/bash
# A/B of environment switches on the default benchmark line (three sessions, codec on): audio-s/s per setting, two rounds.
set -o pipefail
for round in 1 2; do
for v in "X=0" "$@"; do
  env $v timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/abf.log 2>gpurun_out/abf.err || { tail -5 gpurun_out/abf.err; exit 1; }
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/abf.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/abf.log) $(grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/abf.log)"
done
done
