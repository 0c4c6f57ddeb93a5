"""Developer scratch: one S17 B=64 run for rocprofv3."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import synth, Model, LoaderConfig, GenerationRequest, SamplingConfig
shape = sys.argv[1] if len(sys.argv) > 1 else "S17"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg, ccfg, tmpl = synth.shape_set(shape)
m = Model(cfg, None, LoaderConfig(dtype="bf16"))
for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 1, bf16=True):
    m.load_tensor(k, t)
m.finalize()
reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 12 + (b % 16), b), language="english", speaker="speaker1", seed=b) for b in range(B)]
out = m.generate(reqs, SamplingConfig(), frames, use_graph=False)
print(m.last_timing)
m.close()
