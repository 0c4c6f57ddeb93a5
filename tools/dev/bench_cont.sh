#!/bin/bash
# Continuous-batching runs of the benchmark next to the fixed-length default (one GPU).
set -e
out=gpurun_out/cont
mkdir -p $out
run() { tag=$1; shift; timeout -k 10 400 python bench.py --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err; python - <<PY
import json
r=json.load(open("$out/$tag.json"))
print("$tag", r["value"], r["ms_per_step"], r["phase_ms_per_step"], r["ms_per_frame"], r["batches_in_flight"], r["config"]["workload"][-110:])
PY
}
run spread --steps 6 --warmup 3 --length-spread 100:375
run spread_arrival --steps 6 --warmup 3 --length-spread 100:375 --queue-order arrival
run spread_g64 --steps 6 --warmup 3 --length-spread 100:375 --codec-group 64
run spread_12 --steps 12 --warmup 3 --length-spread 100:375
run weak_cont_g64 --steps 6 --warmup 3 --continuous --codec-group 64
run strong --mode strong --steps 2 --warmup 1
run strong_spread --mode strong --steps 2 --warmup 1 --length-spread 100:375
