#!/bin/bash
# Throughput of the multi-session job (no codec) against the number of sessions in flight and of hardware queues:
#   tools/dev/pipe_matrix.sh "<queues> <sessions> <steps>" ...
set -o pipefail
out=gpurun_out
i=0
for v in "$@"; do
    set -- $v
    GPU_MAX_HW_QUEUES=$1 timeout -k 10 300 python bench.py --pipeline $2 --steps $3 --warmup $2 --no-codec --no-cpu-baseline --frames 200 > $out/pipe_$i.json 2>/dev/null || exit 2
    python - "$v" $out/pipe_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_frame"], flush=True)
PY
    i=$((i+1))
done
