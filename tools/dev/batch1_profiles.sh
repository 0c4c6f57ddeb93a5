#!/bin/bash
# Batch 1 (the 0.6B-like and the 1.7B-like shape sets, one stream): bench lines and the kernel-trace statistics of the 0.6B run.
# usage: batch1_profiles.sh <tag>; results under gpurun_out/<tag>_batch1_*
set -o pipefail
tag=${1:-r03}
out=/root/repo/gpurun_out
cd /root/repo || exit 1
args="--batch 1 --pipeline 1 --steps 2 --warmup 1 --frames 125 --no-cpu-baseline"
timeout -k 10 300 python3 bench.py --shape S06 $args > $out/${tag}_batch1_S06.json 2>/dev/null || exit 3
timeout -k 10 300 python3 bench.py --shape S17 $args > $out/${tag}_batch1_S17.json 2>/dev/null || exit 4
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_batch1 -o p -- python3 /root/repo/bench.py --shape S06 $args > $out/prof_${tag}_batch1.log 2>&1 || exit 5
rm -f $out/prof_${tag}_batch1/p_kernel_trace.csv
echo done
