#!/bin/bash
# One GPU round trip while working on the decode kernels: parity tests of the generate path, the single-stream frame time with
# and without the variant under test, and a kernel-trace profile of the single-stream run.
#   tools/dev/gpu_cycle.sh <tag> [ENV=VALUE of the comparison run]
set -o pipefail
tag=${1:-x}
cmp_env=${2:-QTTS_GEMM_STREAM=0}
out=gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_generate.py tests/test_gpu_parity_full_size.py -m gpu -x -q --timeout 300 --timeout-method thread > $out/gpu_tests_$tag.log 2>&1 || { tail -30 $out/gpu_tests_$tag.log; exit 1; }
tail -2 $out/gpu_tests_$tag.log
timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-codec --no-cpu-baseline --frames 200 > $out/b1_$tag.json 2>/dev/null || exit 2
env $cmp_env timeout -k 10 200 python bench.py --steps 2 --warmup 1 --pipeline 1 --no-codec --no-cpu-baseline --frames 200 > $out/b1_${tag}_cmp.json 2>/dev/null || exit 3
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /root/repo/$out/prof_$tag -o p -- python3 /root/repo/bench.py --steps 2 --warmup 1 --pipeline 1 --frames 120 --no-codec --no-cpu-baseline > /root/repo/$out/prof_$tag.log 2>&1
rc=$?
rm -f /root/repo/$out/prof_$tag/p_kernel_trace.csv
exit $rc
