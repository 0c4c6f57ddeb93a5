#!/usr/bin/env python3
"""Side measurements for the benchmark configurations that are not the headline run (one GPU):
  * long-form codec decode: 32 rows x 7500 frames (10 minutes each) -> waveforms, with the reference's chunked policy
    (256 / 32) and with the streaming session (pushes of 256 frames, exact);
  * voice-clone front end: 256 clips of 3-10 s -> mel spectrogram + speaker embedding, then one generate call with the
    x-vectors (batch 64 and all 256 at once, 125 frames).
Prints one JSON line per measurement.  Synthetic weights and inputs."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SEED = 20261111


def build(shape, model_type, talker, codec, speaker):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set(shape, model_type)
    m = Model(cfg, ccfg if codec else None, LoaderConfig(dtype="bf16"))
    keys = (synth.talker_keys(cfg) if talker else []) + (synth.speaker_keys(cfg) if speaker else []) + (synth.codec_keys(ccfg) if codec else [])
    for k, t in synth.iter_weights_torch(keys, SEED, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    return m, cfg, ccfg, tmpl


def long_form():
    m, cfg, ccfg, tmpl = build("S17", "custom_voice", False, True, False)
    B, T = 32, 7500
    codes = np.random.default_rng(0).integers(0, 2048, (B, T, 16)).astype(np.uint32)
    audio_s = B * T * 0.08
    for rep in range(2):
        t0 = time.perf_counter()
        w = m.decode_codes(codes, chunk=-2)
        dt = time.perf_counter() - t0
    print(json.dumps({"workload": "codec decode 32 x 7500 frames, reference chunking 256/32", "audio_s_per_s": round(audio_s / dt, 1),
                      "wall_s": round(dt, 3), "device_ms": round(m.codec_device_ms(), 1), "samples": int(len(w[0]))}), flush=True)
    for rep in range(2):
        t0 = time.perf_counter()
        n = 0
        with m.open_codec_stream(B, 256) as s:
            for a in range(0, T, 256):
                n += s.push(codes[:, a:a + 256]).shape[1]
        dt = time.perf_counter() - t0
    print(json.dumps({"workload": "codec decode 32 x 7500 frames, streaming session, pushes of 256 frames (equals one whole decode)",
                      "audio_s_per_s": round(audio_s / dt, 1), "wall_s": round(dt, 3), "samples": int(n)}), flush=True)
    m.close()


def voice_clone():
    from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synth
    m, cfg, ccfg, tmpl = build("S17", "base", True, False, True)
    rng = np.random.default_rng(1)
    clips = []
    for i in range(256):
        n = int(rng.integers(3 * 24000, 10 * 24000))
        t = np.arange(n) / 24000.0
        clips.append((0.3 * np.sin(2 * np.pi * (110 + i) * t) + 0.05 * rng.standard_normal(n)).astype(np.float32))
    for rep in range(2):
        t0 = time.perf_counter()
        emb = m.extract_speaker_embedding(clips)
        dt = time.perf_counter() - t0
    clip_s = sum(len(c) for c in clips) / 24000.0
    print(json.dumps({"workload": "mel + ECAPA speaker embedding, 256 clips of 3-10 s", "clips_per_s": round(256 / dt, 1),
                      "clip_audio_s_per_s": round(clip_s / dt, 1), "wall_s": round(dt, 3)}), flush=True)
    reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 12 + i % 40, i), language="english", speaker_embed=np.asarray(emb[i]), seed=7 + i)
            for i in range(256)]
    for nb in (64, 256):   # 256 = the whole voice-clone configuration as one batch on one GPU
        for rep in range(2):
            t0 = time.perf_counter()
            out = m.generate(reqs[:nb], SamplingConfig(), 125, min_new_tokens=125)
            dt = time.perf_counter() - t0
        print(json.dumps({"workload": f"x-vector voice clone generate, batch {nb}, 125 frames (fixed length), one stream",
                          "audio_s_per_s": round(sum(out.effective_lengths) * 0.08 / dt, 1), "wall_s": round(dt, 3),
                          "ms_per_frame": round(m.last_timing["frames_ms"] / 124, 3)}), flush=True)
    m.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["long_form", "voice_clone"]
    if "long_form" in which:
        long_form()
    if "voice_clone" in which:
        voice_clone()
