"""Developer scratch: do two half-batches on two streams overlap? (two model replicas, two host threads)"""
import os, sys, time, threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth, GenerationRequest, SamplingConfig

cfg, ccfg, tmpl = synth.shape_set("S17")
NM = int(os.environ.get("NMODELS", "2"))
models = []
for i in range(NM):
    m = Model(cfg, None, LoaderConfig(dtype="bf16"))
    for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 20261111, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    models.append(m)
torch.cuda.empty_cache()
samp = SamplingConfig()
FR = 100

def reqs(n, off):
    return [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 8 + ((b + off) * 37 + 11) % 57, b + off), language="english",
                              speaker="speaker1", seed=42 + b + off) for b in range(n)]

def run(m, r, out, i):
    m.generate(r, samp, FR)
    out[i] = m.last_timing["frames_ms"] / (FR - 1)

for B in (64, 32, 16):
    out = [0]
    run(models[0], reqs(B, 0), out, 0)
    t0 = time.perf_counter(); run(models[0], reqs(B, 0), out, 0); t1 = time.perf_counter()
    print(f"single stream B={B}: {out[0]:.3f} ms/frame wall {t1 - t0:.3f}s", flush=True)
for n in range(2, NM + 1):
    B = int(os.environ.get("BPER", 64 // n))
    out = [0] * n
    for rep in range(2):
        th = [threading.Thread(target=run, args=(models[i], reqs(B, i * B), out, i)) for i in range(n)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        t1 = time.perf_counter()
    print(f"{n} concurrent streams x B={B}: per-stream ms/frame {['%.3f' % o for o in out]} wall {t1 - t0:.3f}s -> effective {1e3 * (t1 - t0) / FR / (n * B / 64):.3f} ms per 64-row frame", flush=True)
for m in models: m.close()
