#!/bin/bash
# HBM-side traffic of the decode frame: FETCH_SIZE and WRITE_SIZE (separate passes) of every kernel of an eager (graph-less)
# single-stream run at two frame counts; tools/pmc_frame_traffic.py turns the difference into bytes per frame.
# usage: pmc_frame.sh alone|shared   (which decode kernel selection the single stream runs: QTTS_TUNE)
set -o pipefail
tag=${1:-alone}
export QTTS_TUNE=$tag
cd /tmp && export TMPDIR=/tmp
out=/root/repo/gpurun_out
for f in 12 22; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -s KILL 280 rocprofv3 --pmc $c --output-format csv -d $out/pmc_frame_${tag}_${c}_$f -o p -- python3 /root/repo/bench.py --no-graph --steps 1 --warmup 0 --pipeline 1 --no-codec --no-cpu-baseline --frames $f > $out/pmc_frame_${tag}_${c}_$f.log 2>&1 || exit 1
    python3 - "$out/pmc_frame_${tag}_${c}_$f/p_counter_collection.csv" "$c" > $out/pmc_frame_${tag}_${c}_$f.sum <<'PY'
import csv, sys, collections
tot = collections.Counter(); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] == sys.argv[2]:
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
        tot[k] += float(r["Counter_Value"]); n[k] += 1
for k, v in tot.most_common():
    print(f"{v:.1f}\t{n[k]}\t{k}")
PY
    rm -f $out/pmc_frame_${tag}_${c}_$f/p_counter_collection.csv
  done
done
