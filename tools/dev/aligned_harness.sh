#!/bin/bash
# Do three streams that read the SAME weights at about the same time share them through the memory-side cache?
# gate|up, q|k|v and a code-predictor shape on three streams, weight copies seven apart (default) against aligned.
set -o pipefail
for s in "stream 64 12288 2048 1 300 1" "stream 64 4096 2048 1 300 0" "frag 64 6144 1024 0 300 1" "frag 64 1024 3072 0 300 2"; do
  for a in 0 1; do
    echo -n "aligned=$a  "
    HARNESS_ALIGNED=$a HARNESS_STREAMS=3 QTTS_TUNE=shared timeout -k 5 60 tools/gemm_harness $s || exit 1
  done
done
