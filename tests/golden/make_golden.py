"""Regenerates tests/golden/t0_golden.npz from the CPU oracle (oracle/, a restatement of the reference's
algorithm; the reference itself is Rust and cannot be built in this environment, so these vectors pin the
oracle's behaviour over time and are the inputs/outputs the GPU parity tests replay).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import SEED, make_oracle  # noqa: E402
from oracle import bind  # noqa: E402
from qwen3_tts_rs_amd import synth  # noqa: E402

GREEDY = dict(do_sample=False, subtalker_do_sample=False, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.05,
              subtalker_temperature=0.9, subtalker_top_k=50, subtalker_top_p=1.0)
SAMPLED = dict(do_sample=True, subtalker_do_sample=True, temperature=0.9, top_k=50, top_p=1.0, repetition_penalty=1.05,
               subtalker_temperature=0.9, subtalker_top_k=50, subtalker_top_p=1.0)


def test_audio(n, seed):
    """Band-limited deterministic test signal: three sines + a slow chirp."""
    t = np.arange(n, dtype=np.float64) / 24000.0
    f = [220.0 + 37.0 * seed, 563.0 + 11.0 * seed, 1371.0 + 5.0 * seed]
    x = 0.3 * np.sin(2 * np.pi * f[0] * t) + 0.2 * np.sin(2 * np.pi * f[1] * t + 0.5) + 0.1 * np.sin(2 * np.pi * f[2] * t + 1.1)
    x += 0.1 * np.sin(2 * np.pi * (300.0 + 400.0 * t) * t)
    return x.astype(np.float32)


def main():
    out = {}
    o, cfg, ccfg, tmpl = make_oracle(bind, "T0", "f32", talker=True, codec=True)
    t = cfg["talker_config"]
    text = synth.synthetic_text_ids(tmpl, 12, 0)
    prefill = bind.resolve_codec_prefill(cfg, "english", "speaker1")
    spk = bind.resolve_speaker_token(cfg, "speaker1")
    out["text_ids"] = text
    # (1) CLI-greedy trajectory, 64 frames
    r = o.generate(text, prefill, GREEDY, 64, speaker_token=spk, trace=True)
    out["greedy_codes"] = r["codes"]
    out["greedy_logits_head"] = r["logits"][:4]
    out["greedy_hidden_head"] = r["hidden"][:4]
    out["greedy_min_margin"] = np.float32(r["min_margin"])
    # (2) EOS: re-declare a code the greedy trajectory first emits at a step >= 8 as the EOS id
    cb0 = r["codes"][:, 0]
    k = next(i for i in range(8, len(cb0)) if cb0[i] not in cb0[:i])
    out["eos_id"] = np.int32(cb0[k])
    out["eos_step"] = np.int32(k)
    cfg2 = {**cfg, "talker_config": {**t, "codec_eos_token_id": int(cb0[k])}}
    o.set_model_config(cfg2)
    r2 = o.generate(text, prefill, GREEDY, 64, speaker_token=spk)
    out["eos_codes"] = r2["codes"]
    out["eos_num_steps"] = np.int32(r2["num_steps"])
    out["eos_effective_len"] = np.int32(r2["effective_len"])
    o.set_model_config(cfg)
    # (3) seeded sampling (seed 42), 64 frames
    r3 = o.generate(text, prefill, SAMPLED, 64, seed=42, speaker_token=spk)
    out["sampled_codes"] = r3["codes"]
    # (4) prompts: custom voice, voice design (instruct, auto language), x-vector clone, ICL
    p, tr = o.build_prompt(text, prefill, speaker_token=spk)
    out["prompt_custom"], out["trailing_custom"] = p, tr
    ins = synth.synthetic_text_ids(tmpl, 6, 7)
    p, tr = o.build_prompt(text, bind.resolve_codec_prefill(cfg, "auto", None), instruct_ids=ins)
    out["instruct_ids"] = ins
    out["prompt_design"] = p
    xvec = synth.make_tensor("golden.xvector", (t["hidden_size"],), "normal:1.0", SEED)
    p, tr = o.build_prompt(text, prefill, speaker_embed=xvec)
    out["xvector"] = xvec
    out["prompt_clone"] = p
    rng = np.random.default_rng(5)
    ref_codes = rng.integers(0, t["code_predictor_config"]["vocab_size"], (9, t["num_code_groups"])).astype(np.uint32)
    ref_codes[:, 0] = rng.integers(0, 256, 9)
    ref_text = synth.synthetic_text_ids(tmpl, 5, 3)
    p, tr = o.build_prompt(text, prefill, speaker_embed=xvec, ref_text_ids=ref_text, ref_codes=ref_codes, non_streaming=False)
    out["ref_codes"], out["ref_text_ids"] = ref_codes, ref_text
    out["prompt_icl"], out["trailing_icl"] = p, tr
    r4 = o.generate(text, prefill, GREEDY, 24, speaker_embed=xvec, ref_text_ids=ref_text, ref_codes=ref_codes, non_streaming=False)
    out["icl_codes"] = r4["codes"]
    # (5) code predictor on the first frame's hidden state: codes + logits of every pass
    cp_codes, cp_logits = o.code_predictor(r["hidden"][0], int(r["codes"][0, 0]), GREEDY)
    out["cp_codes"], out["cp_logits"] = cp_codes, cp_logits
    # (6) codec: full decode (20 frames), chunked 8/3 (33 frames), decode_output policy on 600 frames (> 512 -> 256/32)
    crng = np.random.default_rng(0)
    codes = crng.integers(0, ccfg["decoder_config"]["codebook_size"], (600, ccfg["decoder_config"]["num_quantizers"])).astype(np.uint32)
    out["codec_codes"] = codes
    out["wav_full20"] = o.codec_decode(codes[:20])
    out["wav_chunked33"] = o.codec_decode(codes[:33], mode=1, chunk=8, left_ctx=3)
    out["wav_policy600"] = o.codec_decode(codes, mode=2)
    out["wav_full600_head"] = o.codec_decode(codes[:300])[:4096]
    for s in range(4):
        out[f"codec_stage{s}"] = o.codec_trace(codes[:20], s)
    # (7) mel + speaker encoder (Base model type carries the ECAPA weights)
    ob, cfgb, _, _ = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=False)
    audio = test_audio(24000, 1)
    mel = ob.mel(audio)
    out["mel_1s"] = mel
    out["spk_embed_1s"] = ob.speaker_embed(mel)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "t0_golden.npz"), **out)
    print({k: (v.shape, str(v.dtype)) for k, v in out.items()})


if __name__ == "__main__":
    main()
