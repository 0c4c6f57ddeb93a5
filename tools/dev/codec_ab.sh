#!/bin/bash
# A/B of environment switches on the codec decode alone (64 x 375 frames): device ms per setting, two rounds.
set -o pipefail
for round in 1 2; do
for v in "X=0" "$@"; do
  env $v REPS=3 timeout -k 10 200 python -u tools/dev/dev_codec_prof.py > gpurun_out/cab.log 2>&1 || { tail -5 gpurun_out/cab.log; exit 1; }
  echo "$v $(grep '^decode' gpurun_out/cab.log | tail -1)"
done
done
