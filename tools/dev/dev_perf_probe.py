"""Developer scratch (not collected by pytest): frame-time probe on a large shape set."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import synth, Model, LoaderConfig, GenerationRequest, SamplingConfig

shape = sys.argv[1] if len(sys.argv) > 1 else "S17"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 32
cfg, ccfg, tmpl = synth.shape_set(shape)
m = Model(cfg, None, LoaderConfig(dtype="bf16"))
t0 = time.time()
for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 1, bf16=True):
    m.load_tensor(k, t)
m.finalize()
torch.cuda.synchronize()
print("load", time.time() - t0, flush=True)
wb = synth.weight_bytes_per_frame(cfg)
for B in (64, 1, 8):
    reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 12 + (b % 16), b), language="english", speaker="speaker1", seed=b) for b in range(B)]
    samp = SamplingConfig()
    m.generate(reqs, samp, 4)
    for graph in (True, False):
        t0 = time.time()
        out = m.generate(reqs, samp, frames, use_graph=graph, profile_gemm=not graph)
        dt = time.time() - t0
        tm = m.last_timing
        per = tm["frames_ms"] / max(1, tm["frames"])
        line = f"B={B} graph={graph} wall={dt:.3f}s prefill={tm['prefill_ms']:.1f}ms frame={per:.3f}ms -> {B*0.08/per*1e3:.0f} audio-s/s (gen only); weights/frame={wb['total']/1e9:.2f}GB -> {wb['total']/per/1e6:.0f} GB/s"
        if not graph and tm["gemm_launches"]:
            line += f" | gemm {tm['gemm_ms']/tm['frames']:.3f} ms/frame, {tm['gemm_launches']} launches, {tm['gemm_bytes']/tm['gemm_ms']/1e6:.0f} GB/s in-kernel"
        print(line, flush=True)
m.close()
