"""Error budget of bf16 activation storage on the 1.7B-like talker: first-frame logits of the CPU oracle with individual
intermediate tensors rounded to bf16 (oracle round probe) against the unrounded run, same bf16-rounded weights."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from oracle import bind
from qwen3_tts_rs_amd import synth

shape = sys.argv[1] if len(sys.argv) > 1 else "S17"
cfg, ccfg, tmpl = synth.shape_set(shape)
o = bind.Oracle(fast=True)
gen = torch.Generator().manual_seed(1)
for key, shp, kind in synth.talker_keys(cfg):
    if "text_embedding" in key:   # only a few rows are used: keep it small
        x = torch.randn(tuple(shp), generator=gen) * 0.02 if int(np.prod(shp)) < 5e7 else torch.zeros(tuple(shp)).normal_(0, 1, generator=gen)
    else:
        x = torch.randn(tuple(shp), generator=gen)
    k = kind.split(":")
    if k[0] == "normal": x.mul_(float(k[1]))
    elif k[0] == "fanin": x.mul_(float(k[1]) / np.sqrt(float(np.prod(shp[1:]))))
    elif k[0] == "ones": x.mul_(float(k[1])).add_(1.0)
    elif k[0] == "const": x.mul_(float(k[2])).add_(float(k[1]))
    x = x.to(torch.bfloat16).float()
    o.load_tensor(key, x.numpy())
    del x
o.set_model_config(cfg)
GREEDY = dict(do_sample=False, subtalker_do_sample=False, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.05)
res = {}
for mask in (0, 1, 2, 4, 8, 15, 2 | 0, 1 | 4 | 8):
    bind.lib(True).orc_set_round_probe(mask)
    errs = []
    for u in range(3):
        text = synth.synthetic_text_ids(tmpl, 8 + (u * 37 + 11) % 57, u)
        t0 = time.time()
        r = o.generate(text, bind.resolve_codec_prefill(cfg, "english", "speaker1"), GREEDY, 1,
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), trace=True)
        if mask == 0:
            res[u] = r["logits"][0].copy()
            print("row", u, "logit std", res[u][:2048].std(), "time", time.time() - t0, flush=True)
        else:
            errs.append(float(np.abs(r["logits"][0] - res[u]).max()))
    if mask:
        print(f"mask {mask:2d}: max-abs logit error per row {errs}", flush=True)
