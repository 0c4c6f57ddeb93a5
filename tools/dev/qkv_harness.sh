#!/bin/bash
# qkv-shaped decode GEMM (64 x 4096 x 2048, two activation planes) through the kernels that can run it, one stream and three.
set -o pipefail
for n in 1 3; do
for s in "stream 64 4096 2048 1 300 0" "full_k 64 4096 2048 1 300 0" "k_split 64 4096 2048 1 300 0" "stream 64 12288 2048 1 300 1"; do
  HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $s || exit 1
done
done
