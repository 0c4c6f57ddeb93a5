"""Developer scratch: turn what tools/dev/round_profiles.sh left under gpurun_out/ (prefix <src>) into the committed files under
profiles/ (prefix <dst>): bench lines, kernel-trace statistics with a per-frame summary, GEMM harness output, traffic figures.

    python tools/dev/collect_profiles.py r03a r03"""
import csv, json, os, re, shutil, sys
src, dst = sys.argv[1], sys.argv[2]
G, P = "gpurun_out", "profiles"
for t in ("bench", "single_stream", "strong_n1", "length_spread"):
    f = f"{G}/{src}_{t}.json"
    if os.path.exists(f):
        line = open(f).read().strip().split("\n")[-1]
        json.loads(line)
        open(f"{P}/{dst}_{t}.json", "w").write(line + "\n")
if os.path.exists(f"{G}/{src}_gemm_harness.txt"):
    shutil.copy(f"{G}/{src}_gemm_harness.txt", f"{P}/{dst}_gemm_harness.txt")
for name, cmd, frames in (("bench", "python3 bench.py --no-cpu-baseline (default: 3 sessions x 64 rows, 375 frames, codec on)", None),
                          ("single", "python3 bench.py --steps 2 --warmup 1 --pipeline 1 --frames 120 --no-codec --no-cpu-baseline", None)):
    f = f"{G}/prof_{src}_{name}/p_kernel_stats.csv"
    if not os.path.exists(f):
        continue
    out = f"{P}/{dst}_{'bench' if name == 'bench' else 'single_stream'}_kernel_stats"
    shutil.copy(f, out + ".csv")
    rows = list(csv.DictReader(open(f)))
    # frames of the run = launches of the step-advance kernel (one per decode frame)
    adv = [r for r in rows if "advance_step" in r["Name"]]
    frames = float(adv[0]["Calls"]) if adv else 1.0
    with open(out + "_summary.txt", "w") as o:
        o.write(f"# {dst} - rocprofv3 --kernel-trace --stats -- {cmd}\n")
        o.write(f"# S17 shape set, bf16, one MI355X.  Per-frame columns divide by the {int(frames)} decode frames of the run (launches of advance_step).\n")
        o.write(f"{'kernel':84s} {'calls':>8s} {'total_ms':>9s} {'avg_us':>8s} {'pct':>6s} {'calls/frame':>11s} {'us/frame':>9s}\n")
        for r in rows[:34]:
            n = r["Name"].replace("void ", "").replace("qtts::", "").replace("(anonymous namespace)::", "")
            n = re.split(r"\(", n)[0][:84]
            calls = int(r["Calls"]); tot = float(r["TotalDurationNs"]) / 1e6
            o.write(f"{n:84s} {calls:8d} {tot:9.2f} {float(r['AverageNs']) / 1e3:8.2f} {float(r['Percentage']):6.2f} {calls / frames:11.1f} {tot * 1e3 / frames:9.1f}\n")
for f in ("frame_traffic.json", "codec_traffic.json"):
    shutil.copy(f"bench_data/{f}", f"{P}/{dst}_{f}")
for f in sorted(os.listdir(G)):
    if f.startswith("codec_prof") and f.endswith(".txt") and f == "codec_prof.txt":
        lines = [l for l in open(f"{G}/{f}") if l.startswith("codec conv") or l.startswith("decode")]
        open(f"{P}/{dst}_codec_conv_profile.txt", "w").write(
            f"# {dst} - codec decoder, 64 utterances x 375 frames, bf16, reference-default codec shape: HIP-event time of every conv launch\n"
            "# (QTTS_CODEC_PROFILE=1 REPS=3 python3 tools/dev/dev_codec_prof.py; two groups of utterances per decode; the last repetition is shown).\n"
            "# taps = 8 rows are fused residual units (7-tap conv + SnakeBeta + 1x1 conv + residual, one launch; the last one of a decode includes the output conv),\n# taps = 16 rows two such units in one launch.\n" + "".join(lines[-62:]))
print("ok")
