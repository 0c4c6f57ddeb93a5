"""Developer scratch: sanity of the 1.7B-like shape set at several batch sizes."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth, GenerationRequest, SamplingConfig
cfg, ccfg, tmpl = synth.shape_set(os.environ.get("SHAPE", "S17"))
m = Model(cfg, None, LoaderConfig(dtype="bf16"))
for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 20261111, "cuda:0", True):
    if not torch.isfinite(t.float()).all():
        print("non-finite weight", k)
    m.load_tensor(k, t)
m.finalize()
g = SamplingConfig.cli_greedy()
ref = None
for B in ((64,) if os.environ.get('ONLY64') else (1, 2, 16, 17, 33, 64)):
    texts = [synth.synthetic_text_ids(tmpl, 8 + (7 * b) % 40, b // 2 if os.environ.get('ONLY64') else b) for b in range(B)]
    reqs = [GenerationRequest(text_ids=t, language="english", speaker="speaker1") for t in texts]
    out = m.generate(reqs, g, 4, trace=True, use_graph=False)
    lg = out.logits
    print("B", B, "nan", int(np.isnan(lg).sum()), "inf", int(np.isinf(lg).sum()), "absmax", float(np.nanmax(np.abs(lg))),
          "codes row0", out.codes[0, :2].tolist(), "hidden nan", int(np.isnan(out.hidden).sum()))
    if ref is None:
        ref = lg[0]
    else:
        print("   row0 logits vs B=1 maxabs", float(np.abs(lg[0] - ref).max()))
m.close()
m = Model(cfg, None, LoaderConfig(dtype="bf16"))
for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), 20261111, "cuda:0", True):
    m.load_tensor(k, t)
m.finalize()
B = 64
texts = [synth.synthetic_text_ids(tmpl, 8 + (7 * b) % 40, b // 2) for b in range(B)]
reqs = [GenerationRequest(text_ids=t, language="english", speaker="speaker1") for t in texts]
for steps, trace, graph in [(6, False, True), (6, False, False), (6, True, False), (4, False, False), (4, True, False)]:
    out = m.generate(reqs, g, steps, trace=trace, use_graph=graph)
    print("steps", steps, "trace", trace, "graph", graph, "cb0", out.codes[:4, :, 0].tolist(), "steps out", out.num_steps, out.effective_lengths[:4])
m.close()
