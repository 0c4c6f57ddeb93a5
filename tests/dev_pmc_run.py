"""Developer scratch: eager decode frames on the main thread (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth, GenerationRequest, SamplingConfig
cfg, ccfg, tmpl = synth.shape_set("S17")
m = Model(cfg, None, LoaderConfig(dtype="bf16"))
for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 20261111, "cuda:0", True):
    m.load_tensor(k, t)
m.finalize()
F = int(sys.argv[1])
reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 8 + (b * 37 + 11) % 57, b), language="english", speaker="speaker1", seed=42 + b)
        for b in range(64)]
out = m.generate(reqs, SamplingConfig(), F, use_graph=False)
print("frames", out.num_steps, m.last_timing)
m.close()
