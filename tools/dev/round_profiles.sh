#!/bin/bash
# End-of-round measurements (every GPU step under its own time limit; stops at the first failure):
# PMC traffic of the decode frame, the default bench line, its kernel-trace statistics, the per-shape GEMM harness,
# the single-stream frame and the strong-scaling job on one GPU.  Results land in gpurun_out/.
set -o pipefail
out=/root/repo/gpurun_out
cd /root/repo || exit 1
bash tools/dev/pmc_frame.sh || exit 1
mkdir -p bench_data
python3 tools/pmc_frame_traffic.py $out 12 22 > bench_data/r02_frame_traffic.json || exit 2
cp bench_data/r02_frame_traffic.json $out/r02_frame_traffic.json
timeout -k 10 900 python3 bench.py > $out/r02_bench.json 2> $out/r02_bench.err || exit 3
timeout -k 10 300 python3 bench.py --steps 2 --warmup 1 --pipeline 1 --no-codec --no-cpu-baseline --frames 200 > $out/r02_single_stream.json 2>/dev/null || exit 4
timeout -k 10 600 python3 bench.py --mode strong --utterances 256 --steps 1 --warmup 1 --no-cpu-baseline > $out/r02_strong_n1.json 2> $out/r02_strong_n1.err || exit 5
{
  for s in "stream 64 12288 2048 1 300 1" "full_k 64 12288 2048 1 300 1" "slab 64 12288 2048 1 300" "full_k 64 4096 2048 1 300 0" "slab 64 4096 2048 1 300" \
           "k_split 64 2048 2048 1 300" "slab 64 2048 2048 1 300" "k_split 64 2048 6144 1 300" "slab 64 2048 6144 1 300" \
           "full_k 64 4096 1024 0 300 0" "full_k 64 6144 1024 0 300 1" "full_k 64 2048 1024 0 300 0" "k_split 64 1024 2048 0 300" "k_split 64 1024 3072 0 300"; do
    timeout -k 5 60 tools/gemm_harness $s || exit 6
  done
} > $out/r02_gemm_harness.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_r02_bench -o p -- python3 /root/repo/bench.py --no-cpu-baseline > $out/prof_r02_bench.log 2>&1 || exit 7
rm -f $out/prof_r02_bench/p_kernel_trace.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_r02_single -o p -- python3 /root/repo/bench.py --steps 2 --warmup 1 --pipeline 1 --frames 120 --no-codec --no-cpu-baseline > $out/prof_r02_single.log 2>&1 || exit 8
rm -f $out/prof_r02_single/p_kernel_trace.csv
echo done
