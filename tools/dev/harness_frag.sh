#!/bin/bash
# The fragment-order GEMM (k_gemm_frag.hip) against the kernels it would replace, alone and with three streams replaying the same chain.
set -o pipefail
H=tools/gemm_harness
run() { for n in 1 3; do env $1 HARNESS_STREAMS=$n timeout -k 5 60 $H $2 | sed -e "s/^/[$1] /" || exit 1; done; }
for K in 2048 3072; do
  run "QTTS_TUNE=alone" "k_split 64 1024 $K 0 300"
  run "QTTS_TUNE=shared" "k_split 64 1024 $K 0 300"
  run "QTTS_FRAG_WAVES=4" "frag 64 1024 $K 0 300 2"
  run "QTTS_FRAG_WAVES=8" "frag 64 1024 $K 0 300 2"
done
for N in 4096 6144 2048; do
  run "QTTS_TUNE=alone" "full_k 64 $N 1024 0 300"
  run "QTTS_TUNE=shared" "stream 64 $N 1024 0 300"
  run "QTTS_FRAG_WAVES=4" "frag 64 $N 1024 0 300 0"
done
