#!/bin/bash
# A/B of two builds of the library on the codec decode alone: qwen3-tts-rs_amd/libqtts_hip_old.so against libqtts_hip_new.so
# (copied over libqtts_hip.so in the box's scratch tree), two rounds.
set -o pipefail
d=qwen3-tts-rs_amd
for round in 1 2; do
for v in old new; do
  cp $d/libqtts_hip_$v.so $d/libqtts_hip.so
  REPS=3 timeout -k 10 200 python -u tools/dev/dev_codec_prof.py > gpurun_out/soab.log 2>&1 || { tail -5 gpurun_out/soab.log; exit 1; }
  echo "$v $(grep '^decode' gpurun_out/soab.log | tail -1)"
done
done
