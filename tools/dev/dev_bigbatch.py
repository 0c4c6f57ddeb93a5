"""Developer scratch: decode frame time vs batch size, one stream and two streams."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth, GenerationRequest, SamplingConfig
cfg, ccfg, tmpl = synth.shape_set("S17")
models = []
for i in range(2):
    m = Model(cfg, None, LoaderConfig(dtype="bf16"))
    for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 20261111, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    models.append(m)
torch.cuda.empty_cache()
samp = SamplingConfig()
FR = 80
def reqs(n, off):
    return [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 8 + ((b + off) * 37 + 11) % 57, b + off), language="english",
                              speaker="speaker1", seed=42 + b + off) for b in range(n)]
def run(m, r, out, i):
    m.generate(r, samp, FR)
    out[i] = m.last_timing["frames_ms"] / (FR - 1)
for B in (64, 96, 128, 192, 256):
    out = [0]
    run(models[0], reqs(B, 0), out, 0)
    run(models[0], reqs(B, 0), out, 0)
    print(f"1 stream  B={B:3d}: {out[0]:7.3f} ms/frame -> {out[0] * 64 / B:.3f} ms per 64 rows", flush=True)
for B in (96, 128):
    out = [0, 0]
    for rep in range(2):
        th = [threading.Thread(target=run, args=(models[i], reqs(B, i * B), out, i)) for i in range(2)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        t1 = time.perf_counter()
    print(f"2 streams B={B:3d}: {out[0]:7.3f} ms/frame each -> {max(out) * 64 / (2 * B):.3f} ms per 64 rows", flush=True)
for m in models: m.close()
