"""Error budget of bf16 activation storage in the reference-default codec decoder: waveform SNR of the CPU oracle with
individual tensors rounded to bf16 (oracle round probe) against the unrounded run, same bf16-rounded weights."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import snr_db, SEED
from oracle import bind
from qwen3_tts_rs_amd import synth, round_bf16

cfg, ccfg, tmpl = synth.shape_set("S17")
o = bind.Oracle(fast=True)
for k, a in synth.iter_weights(synth.codec_keys(ccfg), SEED):
    o.load_tensor(k, round_bf16(a))
o.set_codec_config(ccfg)
rng = np.random.default_rng(5)
codes = rng.integers(0, 2048, (12, 16)).astype(np.uint32)
L = bind.lib(True)
L.orc_set_round_probe(0)
t0 = time.time(); ref = o.codec_decode(codes); print("ref", time.time() - t0, "s", flush=True)
names = {8: "rvq out", 9: "pre_conv out", 10: "transformer out", 11: "upsample convT out", 12: "dwconv+LN out", 13: "GELU out",
         14: "ConvNeXt out (residual)", 15: "block snake (convT in)", 16: "act1 snake (conv1 in)", 17: "act2 snake (conv2 in)", 18: "final snake", 19: "block 0 snake", 20: "block 1 snake", 21: "block 2 snake", 22: "block 3 snake"}
for b, n in names.items():
    L.orc_set_round_probe(1 << b)
    print(f"bit {b:2d} {n:28s}: {snr_db(ref, o.codec_decode(codes)):6.1f} dB", flush=True)
allm = sum(1 << b for b in names if b < 19)
for label, mask in (("all", allm), ("all but 16,17", allm & ~(3 << 16)), ("all but 15,16,17", allm & ~(7 << 15)), ("all but 15-18", allm & ~(15 << 15)),
                    ("only 8-14", sum(1 << b for b in range(8, 15))), ("16,17,18 + blocks 1-3", (7 << 16) | (7 << 20)), ("16,17,18 + blocks 2-3", (7 << 16) | (3 << 21))):
    L.orc_set_round_probe(mask)
    print(f"{label:20s}: {snr_db(ref, o.codec_decode(codes)):6.1f} dB", flush=True)
