#!/bin/bash
# sessions in flight x hardware queues on the default line (codec on): usage pipe4.sh "P Q" ...
set -o pipefail
for cfg in "$@"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout -k 10 400 python -u bench.py --pipeline $1 --steps 12 --warmup 4 --no-cpu-baseline > gpurun_out/p4.log 2>gpurun_out/p4.err || { tail -5 gpurun_out/p4.err; exit 1; }
  echo "pipeline $1 queues $2: $(grep -o '"value": [0-9.]*' gpurun_out/p4.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/p4.log) $(grep -o '"phase_ms_per_step": {[^}]*}' gpurun_out/p4.log)"
done
