#!/bin/bash
# codec decode of 64 x 375 frames under several scratch budgets (MB per handle): device ms and free-memory drop
for mb in 49152 24576 12288 6144; do
  echo "== QTTS_CODEC_SCRATCH_MB=$mb"
  QTTS_CODEC_SCRATCH_MB=$mb REPS=3 timeout -k 10 200 python -u tools/dev/dev_codec_prof.py 2>&1 | grep "^decode" | tail -2 || exit 1
done
