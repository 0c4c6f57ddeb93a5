#!/bin/bash
set -o pipefail
for cfg in "64 3" "128 2" "86 3"; do
  set -- $cfg
  timeout -k 10 400 python -u bench.py --mode strong --batch $1 --pipeline $2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/sm.log 2>gpurun_out/sm.err || { tail -3 gpurun_out/sm.err; exit 1; }
  echo "strong, batches of $1 on $2 sessions: $(grep -o '"value": [0-9.]*' gpurun_out/sm.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/sm.log) $(grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/sm.log)"
done
