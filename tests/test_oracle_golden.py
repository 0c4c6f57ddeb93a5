"""The CPU oracle against the committed golden vectors (tests/golden/t0_golden.npz, written by
tests/golden/make_golden.py) and against properties of the reference's algorithm that need no fixture."""
import numpy as np
import pytest

from conftest import make_oracle
from golden.make_golden import GREEDY, SAMPLED, test_audio as make_audio

FTOL = 2e-6  # the oracle is one fixed binary; the slack only covers libm differences between hosts


@pytest.fixture(scope="module")
def ctx(bind, oracle_t0):
    o, cfg, ccfg, tmpl = oracle_t0
    return dict(o=o, cfg=cfg, ccfg=ccfg, tmpl=tmpl, prefill=bind.resolve_codec_prefill(cfg, "english", "speaker1"),
                spk=bind.resolve_speaker_token(cfg, "speaker1"))


def test_greedy_trajectory(ctx, golden):
    r = ctx["o"].generate(golden["text_ids"], ctx["prefill"], GREEDY, 64, speaker_token=ctx["spk"], trace=True)
    assert np.array_equal(r["codes"], golden["greedy_codes"])
    assert np.abs(r["logits"][:4] - golden["greedy_logits_head"]).max() < FTOL * 10
    assert np.abs(r["hidden"][:4] - golden["greedy_hidden_head"]).max() < FTOL * 10
    # every code of the 64 frames stays below the suppressed range (ids >= V - 1024) and inside the codebook
    V = ctx["cfg"]["talker_config"]["vocab_size"]
    assert (r["codes"][:, 0] < V - 1024).all()
    assert r["num_steps"] == 64 and r["effective_len"] == 64


def test_seeded_sampling_trajectory(ctx, golden):
    r = ctx["o"].generate(golden["text_ids"], ctx["prefill"], SAMPLED, 64, seed=42, speaker_token=ctx["spk"])
    assert np.array_equal(r["codes"], golden["sampled_codes"])
    r2 = ctx["o"].generate(golden["text_ids"], ctx["prefill"], SAMPLED, 64, seed=43, speaker_token=ctx["spk"])
    assert not np.array_equal(r2["codes"], golden["sampled_codes"])


def test_eos_stops_generation(ctx, golden):
    cfg = ctx["cfg"]
    cfg2 = {**cfg, "talker_config": {**cfg["talker_config"], "codec_eos_token_id": int(golden["eos_id"])}}
    o = ctx["o"]
    o.set_model_config(cfg2)
    try:
        r = o.generate(golden["text_ids"], ctx["prefill"], GREEDY, 64, speaker_token=ctx["spk"])
    finally:
        o.set_model_config(cfg)
    k = int(golden["eos_step"])
    assert r["num_steps"] == int(golden["eos_num_steps"]) == k + 1
    assert r["effective_len"] == int(golden["eos_effective_len"]) == k
    assert np.array_equal(r["codes"], golden["eos_codes"])
    assert r["codes"][k, 0] == int(golden["eos_id"])
    # before the EOS the trajectory equals the run without an EOS id
    assert np.array_equal(r["codes"][:k], golden["greedy_codes"][:k])


def test_prompts(ctx, bind, golden):
    o, cfg = ctx["o"], ctx["cfg"]
    p, tr = o.build_prompt(golden["text_ids"], ctx["prefill"], speaker_token=ctx["spk"])
    assert p.shape == golden["prompt_custom"].shape and np.abs(p - golden["prompt_custom"]).max() < FTOL
    # non-streaming mode: trailing is the single tts_pad row
    assert tr.shape == (1, cfg["talker_config"]["hidden_size"]) and np.abs(tr - golden["trailing_custom"]).max() < FTOL
    # prompt length: 3 role + codec prefill (4) + speaker + pad/bos bookkeeping + text + eos + bos
    n = len(golden["text_ids"]) - 8
    assert p.shape[0] == 3 + (len(ctx["prefill"]) + 1 + 2 - 1) + (n + 1) + 1
    p, _ = o.build_prompt(golden["text_ids"], bind.resolve_codec_prefill(cfg, "auto", None), instruct_ids=golden["instruct_ids"])
    assert np.abs(p - golden["prompt_design"]).max() < FTOL
    p, _ = o.build_prompt(golden["text_ids"], ctx["prefill"], speaker_embed=golden["xvector"])
    assert np.abs(p - golden["prompt_clone"]).max() < FTOL
    p, tr = o.build_prompt(golden["text_ids"], ctx["prefill"], speaker_embed=golden["xvector"], ref_text_ids=golden["ref_text_ids"],
                           ref_codes=golden["ref_codes"], non_streaming=False)
    assert np.abs(p - golden["prompt_icl"]).max() < FTOL and np.abs(tr - golden["trailing_icl"]).max() < FTOL


def test_icl_generation(ctx, golden):
    r = ctx["o"].generate(golden["text_ids"], ctx["prefill"], GREEDY, 24, speaker_embed=golden["xvector"],
                          ref_text_ids=golden["ref_text_ids"], ref_codes=golden["ref_codes"], non_streaming=False)
    assert np.array_equal(r["codes"], golden["icl_codes"])


def test_code_predictor_prefill_is_unmasked(ctx, bind, golden):
    """nn/code_predictor.rs:455-460 runs the 2-token prefill without a mask: position 0 attends to position 1.
    The golden logits come from that bidirectional prefill; restating the layer stack with a causal mask on the
    same inputs must give different first-pass logits (this test fails if a causal mask sneaks in)."""
    o, cfg = ctx["o"], ctx["cfg"]
    r = o.generate(golden["text_ids"], ctx["prefill"], GREEDY, 1, speaker_token=ctx["spk"], trace=True)
    codes, logits = o.code_predictor(r["hidden"][0], int(r["codes"][0, 0]), GREEDY)
    assert np.array_equal(codes, golden["cp_codes"]) and np.array_equal(codes, r["codes"][0, 1:])
    assert np.abs(logits - golden["cp_logits"]).max() < FTOL * 10
    # independent causal restatement in numpy for pass 0
    from qwen3_tts_rs_amd import synth
    from conftest import SEED
    t = cfg["talker_config"]
    c = t["code_predictor_config"]
    W = {k: a for k, a in synth.iter_weights(synth.talker_keys(cfg), SEED) if "code_predictor" in k or "codec_embedding" in k}
    H, Hc, nh, nkv, d = t["hidden_size"], c["hidden_size"], c["num_attention_heads"], c["num_key_value_heads"], c["head_dim"]

    def rms(x, w, eps=c.get("rms_norm_eps", 1e-6)):
        return x / np.sqrt((x * x).mean(-1, keepdims=True) + eps) * w

    def rope(x, pos):
        half = d // 2
        inv = (1.0 / (c.get("rope_theta", 1e4) ** (np.arange(half) * 2.0 / d))).astype(np.float32)
        f = (pos[:, None] * inv[None]).astype(np.float32)
        co, si = np.cos(f)[:, None, :], np.sin(f)[:, None, :]
        x1, x2 = x[..., :half], x[..., half:]
        return np.concatenate([x1 * co - x2 * si, x1 * si + x2 * co], -1)

    def run(causal):
        x = np.stack([r["hidden"][0], W["talker.model.codec_embedding.weight"][int(r["codes"][0, 0])]]).astype(np.float32)
        if Hc != H:
            x = x @ W["talker.code_predictor.small_to_mtp_projection.weight"].T + W["talker.code_predictor.small_to_mtp_projection.bias"]
        pos = np.arange(2, dtype=np.float32)
        for l in range(c["num_hidden_layers"]):
            P = f"talker.code_predictor.model.layers.{l}."
            xn = rms(x, W[P + "input_layernorm.weight"])
            q = (xn @ W[P + "self_attn.q_proj.weight"].T).reshape(2, nh, d)
            k = (xn @ W[P + "self_attn.k_proj.weight"].T).reshape(2, nkv, d)
            v = (xn @ W[P + "self_attn.v_proj.weight"].T).reshape(2, nkv, d)
            q = rope(rms(q, W[P + "self_attn.q_norm.weight"]), pos)
            k = rope(rms(k, W[P + "self_attn.k_norm.weight"]), pos)
            k, v = np.repeat(k, nh // nkv, 1), np.repeat(v, nh // nkv, 1)
            s = np.einsum("thd,shd->hts", q, k) * d ** -0.5
            if causal:
                s[:, 0, 1] = -np.inf
            p = np.exp(s - s.max(-1, keepdims=True))
            p /= p.sum(-1, keepdims=True)
            a = np.einsum("hts,shd->thd", p, v).reshape(2, nh * d)
            x = x + a @ W[P + "self_attn.o_proj.weight"].T
            xn = rms(x, W[P + "post_attention_layernorm.weight"])
            g = xn @ W[P + "mlp.gate_proj.weight"].T
            u = xn @ W[P + "mlp.up_proj.weight"].T
            x = x + (g / (1 + np.exp(-g)) * u) @ W[P + "mlp.down_proj.weight"].T
        xn = rms(x, W["talker.code_predictor.model.norm.weight"])
        return xn[1] @ W["talker.code_predictor.lm_head.0.weight"].T

    bidir, causal = run(False), run(True)
    assert np.abs(bidir - logits[0]).max() < 1e-4          # the oracle implements the unmasked prefill
    assert np.abs(causal - logits[0]).max() > 1e-3         # and a causal prefill is measurably different


def test_codec_golden(ctx, golden):
    o = ctx["o"]
    codes = golden["codec_codes"]
    w = o.codec_decode(codes[:20])
    assert w.shape == golden["wav_full20"].shape and np.abs(w - golden["wav_full20"]).max() < FTOL
    up = int(np.prod(ctx["ccfg"]["decoder_config"]["upsample_rates"]) * np.prod(ctx["ccfg"]["decoder_config"]["upsampling_ratios"]))
    assert len(w) < 20 * up and np.abs(w).max() <= 1.0   # every kernel == 2 * stride transposed conv trims both ends
    w = o.codec_decode(codes[:33], mode=1, chunk=8, left_ctx=3)
    assert np.abs(w - golden["wav_chunked33"]).max() < FTOL
    for s in range(4):
        assert np.abs(o.codec_trace(codes[:20], s) - golden[f"codec_stage{s}"]).max() < FTOL * 10


def test_codec_policy_chunks_long_sequences(ctx, golden):
    """Model::decode_output: more than 512 frames => chunked_decode(256, 32), which differs from a full decode."""
    o = ctx["o"]
    codes = golden["codec_codes"]
    w = o.codec_decode(codes, mode=2)
    assert w.shape == golden["wav_policy600"].shape and np.abs(w - golden["wav_policy600"]).max() < FTOL
    w2 = o.codec_decode(codes, mode=1, chunk=256, left_ctx=32)
    assert np.array_equal(w, w2)
    full_head = o.codec_decode(codes[:300])[:4096]
    assert np.abs(full_head - golden["wav_full600_head"]).max() < FTOL
    # the first chunk has no left context, so its samples equal a standalone decode of the first 256 frames
    first = o.codec_decode(codes[:256])
    assert np.array_equal(w[:len(first)], first)
    # causality: a decode of the first 300 frames agrees with the 256-frame decode wherever both are defined
    assert np.abs(full_head[:2048] - first[:2048]).max() < 1e-5


def test_mel_and_speaker_embedding(bind, golden):
    ob, cfgb, _, _ = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=False)
    mel = ob.mel(make_audio(24000, 1))
    assert mel.shape == golden["mel_1s"].shape and np.abs(mel - golden["mel_1s"]).max() < 1e-4
    e = ob.speaker_embed(mel)
    assert np.abs(e - golden["spk_embed_1s"]).max() < 1e-4


def test_encoder_oracle_matches_an_independent_mimi_implementation(bind):
    """The codec-encoder restatement against codes computed by transformers' MimiModel on the same synthetic weights and
    clips (tests/golden/make_mimi_golden.py): an implementation of the same published architecture that shares no code with
    the oracle.  The generator follows the reference in the two places where it departs from Mimi (zero-padded downsample
    conv, no partial last frame); with those, all 34 frames x 16 codebooks are equal."""
    import os
    from conftest import SEED
    from qwen3_tts_rs_amd import synth
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mimi_encoder_golden.npz"))
    o = bind.Oracle()
    for k, a in synth.iter_weights(synth.encoder_keys(), SEED):
        o.load_tensor(k, a)
    for name in ("a", "b"):
        clip = make_audio(int(g["clip_" + name + "_len"]), int(g["clip_" + name + "_seed"]))
        ref = g["codes_" + name]
        got = o.encode_audio(clip, 16)
        assert got.shape == ref.shape and len(ref) == len(clip) // 1920
        assert np.array_equal(got, ref), name


def test_codec_decoder_oracle_matches_an_independent_code2wav_implementation(bind):
    """The codec-decoder restatement against waveforms computed by transformers' Qwen3-Omni Code2Wav modules (transformer,
    upsampling, vocoder) on the same synthetic weights, with the front end the two models do not share written in plain torch
    (tests/golden/make_code2wav_golden.py).  North-star bar: waveform SNR >= 40 dB; the two agree to ~110 dB."""
    import os
    from conftest import snr_db
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "code2wav_golden.npz"))
    o, cfg, ccfg, tmpl = make_oracle(bind, "T0", "f32", talker=False, codec=True)
    for name in ("a", "b"):
        ref = g["wav_" + name]
        w = o.codec_decode(g["codes_" + name])
        assert len(w) == len(ref)
        assert snr_db(ref, w) >= 90.0 and np.abs(w - ref).max() < 1e-4, name


def test_talker_and_code_predictor_oracle_match_an_independent_qwen3_implementation(bind):
    """Talker forward (hidden states, logits) and the greedy code-predictor loop against transformers' Qwen3Model loaded with
    the same synthetic weights (tests/golden/make_qwen3_golden.py).  The oracle walks the reference's interleaved multimodal
    RoPE with equal position rows, the independent model plain RoPE: their agreement is the equivalence the GPU path relies
    on.  The code predictor's first pass is the reference's unmasked two-token prefill."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qwen3_golden.npz"))
    o, cfg, ccfg, tmpl = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    hid, log = o.talker_forward(g["talker_embeds"])
    assert np.abs(hid - g["talker_hidden"]).max() < 2e-5
    assert np.abs(log - g["talker_logits"]).max() < 2e-5
    codes, logits = o.code_predictor(g["talker_hidden"][-1], int(g["cp_cb0"]), GREEDY)
    assert np.array_equal(codes, g["cp_codes"])
    assert np.abs(logits - g["cp_logits"]).max() < 2e-5


def test_mel_oracle_matches_torch_stft(bind):
    """mel_spectrogram (audio/mel.rs:81-158) against the same pipeline written with torch.stft: reflect padding of 384,
    periodic Hann window of 1024, hop 256, magnitude sqrt(re^2 + im^2 + 1e-9), mel filterbank, log(clamp(., 1e-5))."""
    import torch
    o = bind.Oracle()
    audio = make_audio(30000, 9)
    mel = o.mel(audio)
    x = torch.nn.functional.pad(torch.from_numpy(audio)[None, None], (384, 384), mode="reflect")[0]
    spec = torch.stft(x, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024, periodic=True), center=False,
                      return_complex=True)[0]
    mag = torch.sqrt(spec.real ** 2 + spec.imag ** 2 + 1e-9)
    import ctypes as C
    fbn = np.zeros((128, 513), np.float32)                                    # pinned against librosa in test_oracle_kats
    bind.lib().orc_mel_filterbank(1024, 128, 24000, C.c_double(0.0), C.c_double(12000.0), fbn.ctypes.data_as(C.c_void_p))
    fb = torch.from_numpy(fbn)
    ref = torch.log(torch.clamp(fb @ mag, min=1e-5)).T.numpy()
    assert mel.shape == ref.shape
    # compared before the logarithm: near the 1e-5 floor the log turns 1e-8 of rounding into 1e-3
    assert np.abs(np.exp(mel) - np.exp(ref)).max() < 1e-5 * np.exp(ref).max()


def test_speaker_encoder_oracle_matches_an_independent_ecapa_implementation(bind):
    """The ECAPA-TDNN restatement (nn/speaker_encoder.rs) against embeddings computed by transformers' ECAPA_TimeDelayNet
    (the Qwen2.5-Omni speaker encoder: same blocks, same weight keys) on the same synthetic weights and mel inputs of two
    lengths (tests/golden/make_ecapa_golden.py).  The two agree to ~2e-6 on values of magnitude 5."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ecapa_golden.npz"))
    ob, _, _, _ = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=False)
    for name in ("a", "b"):
        e = ob.speaker_embed(g["mel_" + name])
        assert e.shape == g["emb_" + name].shape and np.abs(e - g["emb_" + name]).max() < 2e-5, name
