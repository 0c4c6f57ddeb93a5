#!/bin/bash
# End-of-round measurements (every GPU step under its own time limit; stops at the first failure): the default bench line, the
# single-stream frame, the strong-scaling job and a length-spread job on one GPU, the per-shape GEMM harness, and the kernel-trace
# statistics of the default and the single-stream run.  Results land in gpurun_out/ under the prefix given as $1 (default r03);
# tools/dev/pmc_frame.sh / pmc_codec.sh collect the counter passes separately.
set -o pipefail
tag=${1:-r03}
out=/root/repo/gpurun_out
cd /root/repo || exit 1
timeout -k 10 900 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || exit 3
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --pipeline 1 --no-codec --no-cpu-baseline --frames 200 > $out/${tag}_single_stream.json 2>/dev/null || exit 4
timeout -k 10 600 python3 bench.py --mode strong --utterances 256 --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_strong_n1.json 2> $out/${tag}_strong_n1.err || exit 5
timeout -k 10 600 python3 bench.py --length-spread 100:375 --steps 12 --warmup 3 --no-cpu-baseline > $out/${tag}_length_spread.json 2> $out/${tag}_length_spread.err || exit 5
{
  for s in "stream 64 12288 2048 1 300 1" "full_k 64 12288 2048 1 300 1" "full_k 64 4096 2048 1 300 0" \
           "k_split 64 2048 2048 1 300" "k_split 64 2048 6144 1 300" \
           "frag 64 4096 1024 0 300 0" "full_k 64 4096 1024 0 300 0" "frag 64 6144 1024 0 300 1" "full_k 64 6144 1024 0 300 1" \
           "frag 64 1024 2048 0 300 2" "k_split 64 1024 2048 0 300" "frag 64 1024 3072 0 300 2" "k_split 64 1024 3072 0 300" "frag 64 2048 1024 0 300 0"; do
    timeout -k 5 60 tools/gemm_harness $s || exit 6
  done
} > $out/${tag}_gemm_harness.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_bench -o p -- python3 /root/repo/bench.py --no-cpu-baseline > $out/prof_${tag}_bench.log 2>&1 || exit 7
rm -f $out/prof_${tag}_bench/p_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag}_single -o p -- python3 /root/repo/bench.py --steps 2 --warmup 1 --pipeline 1 --frames 120 --no-codec --no-cpu-baseline > $out/prof_${tag}_single.log 2>&1 || exit 8
rm -f $out/prof_${tag}_single/p_kernel_trace.csv
echo done
