"""Per-shape GEMM time of an eager, HIP-event-timed decode run (S17 bf16, batch 64): launches, average microseconds."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = bench.parse()
args.no_codec = True
m, cfg, ccfg, tmpl = bench.build_model(args, 0)
from qwen3_tts_rs_amd import SamplingConfig
reqs = bench.make_requests(args, tmpl, 0)
frames = 24
m.generate(reqs, SamplingConfig(), frames, use_graph=False, profile_gemm=True, min_new_tokens=frames)
m.generate(reqs, SamplingConfig(), frames, use_graph=False, profile_gemm=True, min_new_tokens=frames)
tot = 0
for p in sorted(m.gemm_profile(), key=lambda p: -p["total_us"]):
    if p["M"] in (64, 128):
        print(f"M={p['M']:4d} N={p['N']:6d} K={p['K']:5d} launches/frame {p['launches'] / frames:6.1f} avg {p['total_us'] / p['launches']:7.2f} us  per frame {p['total_us'] / frames:8.1f} us")
        tot += p["total_us"] / frames
print("GEMM us per frame", round(tot, 1), " frame ms", m.last_timing["frames_ms"] / (frames - 1))
m.close()
