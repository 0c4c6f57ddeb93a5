#!/bin/bash
# A/B of two builds of the library on one box: qwen3-tts-rs_amd/libqtts_hip_old.so against libqtts_hip_new.so (copied over
# libqtts_hip.so in the box's scratch tree), two rounds.  usage: so_ab.sh codec | frame | full
set -o pipefail
d=qwen3-tts-rs_amd
for round in 1 2; do
for v in old new; do
  cp $d/libqtts_hip_$v.so $d/libqtts_hip.so
  case "${1:-codec}" in
    codec) REPS=3 timeout -k 10 200 python -u tools/dev/dev_codec_prof.py > gpurun_out/soab.log 2>&1 || { tail -5 gpurun_out/soab.log; exit 1; }
           echo "$v $(grep '^decode' gpurun_out/soab.log | tail -1)" ;;
    frame) timeout -k 10 200 python -u bench.py --steps 2 --warmup 1 --pipeline 1 --frames 200 --no-codec --no-cpu-baseline > gpurun_out/soab.log 2>&1 || { tail -5 gpurun_out/soab.log; exit 1; }
           echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/soab.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/soab.log)" ;;
    full)  timeout -k 10 300 python -u bench.py --no-cpu-baseline > gpurun_out/soab.log 2>gpurun_out/soab.err || { tail -5 gpurun_out/soab.err; exit 1; }
           echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/soab.log | head -1) $(grep -o '"ms_per_frame": [0-9.]*' gpurun_out/soab.log)" ;;
  esac
done
done
