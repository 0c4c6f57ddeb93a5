"""Developer scratch: codec decoder at benchmark size (64 utterances x 375 frames)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth
cfg, ccfg, tmpl = synth.shape_set("S17")
dtype = os.environ.get("DTYPE", "bf16")
m = Model(cfg, ccfg, LoaderConfig(dtype=dtype))
for k, t in synth.iter_weights_torch(synth.codec_keys(ccfg), 20261111, "cuda:0", dtype == "bf16"):
    m.load_tensor(k, t)
m.finalize()
B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 375))
codes = np.random.default_rng(0).integers(0, 2048, (B, T, 16)).astype(np.uint32)
for rep in range(int(os.environ.get("REPS", 2))):
    t0 = time.perf_counter()
    w = m.decode_codes(codes)
    t1 = time.perf_counter()
    ms = m.codec_device_ms()
    fl = synth.codec_flops_per_frame(ccfg) * B * T
    print(f"decode {B}x{T}: wall {t1 - t0:.3f}s device {ms:.1f} ms -> {fl / ms / 1e9:.1f} TFLOP/s, {B * T * 0.08 / (t1 - t0):.0f} audio-s/s", flush=True)
m.close()
