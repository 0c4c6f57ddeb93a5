#!/usr/bin/env python3
"""Golden speaker embeddings from an independent public implementation of the ECAPA-TDNN speaker encoder.

The reference's speaker encoder (qwen_tts/src/nn/speaker_encoder.rs:44-50,136,203-204,252-267,359-364,489-538) has the
layout and the weight keys of the ECAPA_TimeDelayNet that `transformers` (installed in the build container) ships with
Qwen2.5-Omni's token2wav: reflect-padded "same" TDNN convs, Res2Net blocks with scale 8, squeeze-excitation, multi-layer
feature aggregation, attentive statistics pooling, a 1x1 output conv.  That module is loaded with the synthetic T0/ET0
speaker-encoder weights and run on seeded log-mel-shaped inputs of two lengths.

    python tests/golden/make_ecapa_golden.py          # writes tests/golden/ecapa_golden.npz
"""
import os
import sys
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
SEED = 20261111


def main():
    import torch
    from transformers.models.qwen2_5_omni.modeling_qwen2_5_omni import ECAPA_TimeDelayNet
    from qwen3_tts_rs_amd import synth
    cfg, _, _ = synth.shape_set("T0", "base")
    s = cfg["speaker_encoder_config"]
    net = ECAPA_TimeDelayNet(SimpleNamespace(**s)).eval()
    W = {k[len("speaker_encoder."):]: torch.from_numpy(a) for k, a in synth.iter_weights(synth.speaker_keys(cfg), SEED)}
    sd = net.state_dict()
    assert set(sd) == set(W), (sorted(set(sd) ^ set(W)))
    for k in sd:
        assert tuple(sd[k].shape) == tuple(W[k].shape), (k, sd[k].shape, W[k].shape)
        sd[k].copy_(W[k])
    net.load_state_dict(sd)
    rng = np.random.default_rng(23)
    out = {"loaded_tensors": np.int64(len(sd))}
    for name, T in (("a", 94), ("b", 311)):
        # log-mel-like values: the range ln(1e-5) .. ~2 of audio/mel.rs:45-48
        mel = (rng.standard_normal((T, s["mel_dim"])) * 2.0 - 4.0).clip(-11.5, 2.0).astype(np.float32)
        with torch.no_grad():
            e = net(torch.from_numpy(mel)[None])[0].numpy()
        out["mel_" + name] = mel
        out["emb_" + name] = e.astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "ecapa_golden.npz"), **out)
    print({k: getattr(v, "shape", v) for k, v in out.items()})


if __name__ == "__main__":
    main()
