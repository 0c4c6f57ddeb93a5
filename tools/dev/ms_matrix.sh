#!/bin/bash
# Throughput of the default three-session job (no codec) under a list of environment variants:
#   tools/dev/ms_matrix.sh <tag> "A=1" "B=2 C=3" ...     ("-" = no variable)
set -o pipefail
tag=$1; shift
out=gpurun_out
i=0
for v in "$@"; do
    [ "$v" = "-" ] && v="QTTS_NOP=1"
    env $v timeout -k 10 200 python bench.py --steps 6 --warmup 3 --no-codec --no-cpu-baseline --frames 200 > $out/ms_${tag}_$i.json 2>/dev/null || exit 2
    python - "$v" $out/ms_${tag}_$i.json <<'PY'
import json, sys
d = json.load(open(sys.argv[2]))
print(sys.argv[1], d["value"], d["ms_per_frame"], flush=True)
PY
    i=$((i+1))
done
