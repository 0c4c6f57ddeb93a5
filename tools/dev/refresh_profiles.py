"""Developer scratch: turn the files a profiling run left under gpurun_out/ into the committed summaries under profiles/."""
import csv, json, re, shutil, sys
LABEL = sys.argv[1] if len(sys.argv) > 1 else "final"   # heading and file prefix of the summaries, e.g. "r01"
rows = list(csv.DictReader(open('gpurun_out/p_bench_kernel_stats.csv')))
frames = 720.0
with open('profiles/r01_final_bench_kernel_stats_summary.txt', 'w') as f:
    p = lambda *a: print(*a, file=f)
    p(f"# {LABEL} - rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 3 --warmup 3 --frames 120 --no-cpu-baseline`")
    p("# (S17 shape set, bf16, batch 64 per step, 3 batches in flight on 3 streams, one MI355X; same shape as the default run, fewer frames).")
    p("# Per-frame columns divide by the 720 decode frames of the 6 steps (the isolated codec decode of the roofline leg adds a little).")
    p("# Batch-1 and codec-only tables: r01_batch1_S06_kernel_stats_summary.txt, r01_codec_conv_profile.txt.")
    p(f"{'kernel':76s} {'calls':>8s} {'total_ms':>9s} {'avg_us':>9s} {'pct':>6s} {'calls/frame':>11s}")
    for r in rows[:28]:
        n = r['Name'].replace('void ', '').replace('qtts::', '').replace('(anonymous namespace)::', '')
        n = re.split(r'\(', n)[0][:76]
        calls = int(r['Calls']); tot = float(r['TotalDurationNs']) / 1e6
        p(f"{n:76s} {calls:8d} {tot:9.2f} {float(r['AverageNs'])/1e3:9.2f} {float(r['Percentage']):6.2f} {calls/frames:11.1f}")
shutil.copy('gpurun_out/p_bench_kernel_stats.csv', 'profiles/r01_final_bench_kernel_stats.csv')
shutil.copy('gpurun_out/p_codec_kernel_stats.csv', 'profiles/r01_codec_kernel_stats.csv')
with open('profiles/r01_codec_conv_profile.txt', 'w') as f:
    f.write(f"# {LABEL} - codec decoder, 64 utterances x 375 frames, bf16, default codec shape: HIP-event time of every conv launch\n"
            "# (QTTS_CODEC_PROFILE=1 python3 tools/dev/dev_codec_prof.py; two groups of utterances per decode; the second repetition is shown).\n"
            "# taps = 8 rows are fused residual units (7-tap conv + SnakeBeta + 1x1 conv + residual, one launch); 'in+out' = (input + output channels) x rows x 2 B / time.\n"
            "# History of the same decode: 868 ms (first version) -> 468 (cheaper epilogue math) -> 361 (DMA tile load, batched epilogue loads) -> 335 (fused units) -> 327.\n")
    lines = [l for l in open('gpurun_out/codec_stage_final.log') if l.startswith('codec conv') or l.startswith('decode')]
    f.writelines(lines[-50:])
line = open('gpurun_out/bench_final.log').read().strip().split('\n')[-1]
json.loads(line)
open('profiles/r01_final_bench_default.json', 'w').write(line + '\n')
d = json.loads(line)
print(d['value'], d['roofline']['achieved'], d['roofline']['frac'], d['roofline_codec']['achieved'], d['roofline_codec']['device_ms_isolated'], d['p50_first_audio_ms'], d['p50_full_utterance_ms'], d['cpu_baseline']['value'])
