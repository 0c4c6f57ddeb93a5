#!/bin/bash
# Per-kernel throughput with one and with three streams replaying the same chain of launches (tools/gemm_harness, HARNESS_STREAMS).
set -o pipefail
for s in "k_split 64 1024 2048 0 300" "k_split 64 1024 3072 0 300" "full_k 64 4096 1024 0 300 0" "full_k 64 6144 1024 0 300 1" "full_k 64 2048 1024 0 300 0" \
         "stream 64 12288 2048 1 300 1" "full_k 64 4096 2048 1 300 0" "slab 64 2048 2048 1 300" "slab 64 2048 6144 1 300"; do
  for n in 1 3; do
    HARNESS_STREAMS=$n timeout -k 5 60 tools/gemm_harness $s || exit 1
  done
done
