"""GPU parity on the configuration corners of the generate path: long prompts (chunked prefill), sliding-window talker
attention, attention biases, top-p sampling, disabled repetition penalty, mixed greedy / sampled stages, single-frame
runs.  fp32, token-exact against the CPU oracle."""
import copy

import json

import numpy as np
import pytest

from conftest import SEED, make_oracle

pytestmark = pytest.mark.gpu


def _pair(bind, mutate=None, shape="T0"):
    """(GPU model, oracle, cfg, template) for a shape set whose talker config was edited by `mutate`."""
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set(shape)
    cfg = copy.deepcopy(cfg)
    if mutate:
        mutate(cfg)
    keys = synth.talker_keys(cfg)
    m = Model(cfg, None, LoaderConfig(dtype="f32"))
    o = bind.Oracle()
    for k, a in synth.iter_weights(keys, SEED):
        m.load_tensor(k, a)
        o.load_tensor(k, a)
    m.finalize()
    o.set_model_config(cfg)
    return m, o, cfg, tmpl


def _check(bind, m, o, cfg, texts, sampling, steps, seeds=None, **okw):
    from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig
    sc = SamplingConfig(**{k: sampling[k] for k in SamplingConfig().__dict__ if k in sampling})
    reqs = [GenerationRequest(text_ids=t, language="english", speaker="speaker1", seed=None if seeds is None else seeds[i])
            for i, t in enumerate(texts)]
    out = m.generate(reqs, sc, steps)
    for i, t in enumerate(texts):
        r = o.generate(t, bind.resolve_codec_prefill(cfg, "english", "speaker1"), sampling, steps, seed=0 if seeds is None else seeds[i],
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), **okw)
        assert np.array_equal(r["codes"], out.codes[i][:r["num_steps"]]), f"utterance {i}"
    return out


BASE = dict(do_sample=False, subtalker_do_sample=False, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.05,
            subtalker_temperature=0.9, subtalker_top_k=50, subtalker_top_p=1.0)


def test_long_prompt_spans_several_prefill_chunks(bind):
    from qwen3_tts_rs_amd import synth
    m, o, cfg, tmpl = _pair(bind)
    try:
        texts = [synth.synthetic_text_ids(tmpl, n, i) for i, n in enumerate([3700, 40, 700])]   # 4400+ prompt positions in one batch: more than one prefill pass of 4096
        _check(bind, m, o, cfg, texts, BASE, 6)
    finally:
        m.close()


def test_sliding_window_talker(bind):
    """A non-null `sliding_window` in the talker config limits every query to its last W keys (talker_config.rs:360-362)."""
    from qwen3_tts_rs_amd import synth

    def mut(cfg):
        cfg["talker_config"]["sliding_window"] = 16

    m, o, cfg, tmpl = _pair(bind, mut)
    try:
        texts = [synth.synthetic_text_ids(tmpl, n, i) for i, n in enumerate([30, 12])]
        out = _check(bind, m, o, cfg, texts, BASE, 24)
        m2, o2, cfg2, _ = _pair(bind)
        try:
            ref = _check(bind, m2, o2, cfg2, texts, BASE, 24)
            assert not np.array_equal(ref.codes, out.codes)       # the window really changes the result
        finally:
            m2.close()
    finally:
        m.close()


def test_code_predictor_layers_typed_sliding_attention(bind):
    """`layer_types` may mark code-predictor layers "sliding_attention" (nn/attention/standard.rs:28-39).  The code predictor never
    sees more than num_code_groups + 1 positions, so with the default window (4096) such a layer is a full-attention layer: same
    tokens as without the marking, against the oracle too.  A window shorter than the context is refused with a clear message."""
    from qwen3_tts_rs_amd import LoaderConfig, Model, QttsError, synth

    def mut(cfg):
        cp = cfg["talker_config"]["code_predictor_config"]
        cp["layer_types"] = ["full_attention" if i % 2 else "sliding_attention" for i in range(cp["num_hidden_layers"])]
        cp["sliding_window"] = 4096

    m, o, cfg, tmpl = _pair(bind, mut)
    try:
        texts = [synth.synthetic_text_ids(tmpl, n, i) for i, n in enumerate([9, 17])]
        out = _check(bind, m, o, cfg, texts, BASE, 12)
        m2, o2, cfg2, _ = _pair(bind)
        try:
            ref = _check(bind, m2, o2, cfg2, texts, BASE, 12)
            assert np.array_equal(ref.codes, out.codes)
        finally:
            m2.close()
        short = json.loads(json.dumps(cfg))
        short["talker_config"]["code_predictor_config"]["sliding_window"] = 4
        with pytest.raises(QttsError, match="sliding"):
            Model(short, None, LoaderConfig(dtype="f32"))
    finally:
        m.close()


def test_attention_bias(bind):
    from qwen3_tts_rs_amd import synth

    def mut(cfg):
        cfg["talker_config"]["attention_bias"] = True
        cfg["talker_config"]["code_predictor_config"]["attention_bias"] = True

    m, o, cfg, tmpl = _pair(bind, mut)
    try:
        _check(bind, m, o, cfg, [synth.synthetic_text_ids(tmpl, 14, 2)], BASE, 10)
    finally:
        m.close()


@pytest.mark.parametrize("name,delta", [
    ("top_p", dict(do_sample=True, subtalker_do_sample=True, top_p=0.8, subtalker_top_p=0.7, temperature=0.9)),
    ("no_top_k", dict(do_sample=True, subtalker_do_sample=True, top_k=0, subtalker_top_k=0)),
    ("no_penalty_hot", dict(do_sample=True, subtalker_do_sample=True, repetition_penalty=1.0, temperature=1.3, subtalker_temperature=1.0)),
    ("mixed_greedy_talker", dict(do_sample=False, subtalker_do_sample=True)),
    ("mixed_greedy_subtalker", dict(do_sample=True, subtalker_do_sample=False)),
    ("strong_penalty", dict(repetition_penalty=1.8)),
])
def test_sampling_variants(bind, name, delta):
    from qwen3_tts_rs_amd import synth
    m, o, cfg, tmpl = _pair(bind)
    try:
        texts = [synth.synthetic_text_ids(tmpl, 10 + 3 * i, i) for i in range(3)]
        _check(bind, m, o, cfg, texts, dict(BASE, **delta), 32, seeds=[42, 7, 2026])
    finally:
        m.close()


@pytest.mark.parametrize("delta", [dict(do_sample=True, subtalker_do_sample=True), dict(do_sample=False, subtalker_do_sample=True),
                                   dict(do_sample=True, subtalker_do_sample=False)], ids=["both", "subtalker-only", "talker-only"])
def test_shared_draw_stream_follows_the_reference_order(bind, delta):
    """shared_rng_stream: the rows of a batch draw from ONE MT19937 stream in the order of the reference's global generator
    (nn/mt_rng.rs:22: per frame rows 0..n-1 at the talker, generation.rs:1055-1060, then rows 0..n-1 at every residual codebook,
    code_predictor.rs:468-472,517-522).  Row b of n therefore owns elements b, b + n, ... of the stream, which is how the
    oracle reproduces each row on its own."""
    from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synth
    m, o, cfg, tmpl = _pair(bind)
    try:
        sampling = dict(BASE, **delta)
        sc = SamplingConfig(**{k: sampling[k] for k in SamplingConfig().__dict__ if k in sampling})
        texts = [synth.synthetic_text_ids(tmpl, 10 + 3 * i, i) for i in range(3)]
        reqs = [GenerationRequest(text_ids=t, language="english", speaker="speaker1", seed=42) for t in texts]
        out = m.generate(reqs, sc, 24, shared_rng_stream=True)
        for i, t in enumerate(texts):
            r = o.generate(t, bind.resolve_codec_prefill(cfg, "english", "speaker1"), sampling, 24, seed=42,
                           speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), rng_stride=len(texts), rng_offset=i)
            assert r["num_steps"] >= 2 and np.array_equal(r["codes"], out.codes[i][:r["num_steps"]]), f"utterance {i}"
        own = m.generate(reqs, sc, 24)      # one stream per request: rows 1, 2 see other draws
        assert not np.array_equal(own.codes[1], out.codes[1])
        with pytest.raises(Exception):
            m.generate(reqs, sc, 8, slots=2, shared_rng_stream=True)
    finally:
        m.close()


def test_single_frame_and_minimal_text(bind):
    from qwen3_tts_rs_amd import synth
    m, o, cfg, tmpl = _pair(bind)
    try:
        texts = [synth.synthetic_text_ids(tmpl, 1, 0), synth.synthetic_text_ids(tmpl, 12, 1)]
        _check(bind, m, o, cfg, texts, BASE, 1)
        _check(bind, m, o, cfg, texts, BASE, 2)
        # nine ids is the shortest prompt of the non-streaming branch; eight takes the streaming layout (nn/generation.rs:699)
        _check(bind, m, o, cfg, [texts[0][[0, 1, 2, 3, -5, -4, -3, -2, -1]][:8]], BASE, 3)
    finally:
        m.close()


def test_mid_size_codec_f32(bind):
    """Mid-size talker shapes share the tiny codec; its f32 decode of generated codes stays above the waveform bar."""
    from conftest import snr_db
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("M0", dtype="f32", codec=True, talker=False)
    o, _, _, _ = make_oracle(bind, "M0", "f32", talker=False, codec=True)
    try:
        codes = np.random.default_rng(9).integers(0, 256, (2, 50, 4)).astype(np.uint32)
        wavs = m.decode_codes(codes, [50, 17])
        for b, L in enumerate([50, 17]):
            ref = o.codec_decode(codes[b, :L])
            assert len(ref) == len(wavs[b]) and snr_db(ref, wavs[b]) >= 40.0
    finally:
        m.close()


def test_string_front_ends_tokenise_then_generate(bind, tmp_path):
    """generate_*_from_text (model/generate.rs:680-794): text -> chat template -> ids -> the *_full path.  The tokenizer is a
    toy byte-level BPE written in the tokenizer.json format (its ids stay inside T0's small text vocabulary)."""
    from test_text_front_end import _toy_tokenizer
    from qwen3_tts_rs_amd import QttsError
    from qwen3_tts_rs_amd.text import TextProcessor
    m, o, cfg, tmpl = _pair(bind)
    try:
        with pytest.raises(QttsError, match="No text processor loaded"):
            m.generate_custom_voice_from_text("hello world", "speaker1", "english")
        _toy_tokenizer(str(tmp_path / "tokenizer.json"))
        m.text_processor = TextProcessor.from_pretrained(str(tmp_path))
        opts = dict(max_new_tokens=6, do_sample=False, subtalker_do_sample=False, temperature=1.0)
        ids = m.tokenize_text("hello world speak slowly")
        assert max(ids) < cfg["talker_config"]["text_vocab_size"] and len(ids) > 8
        a = m.generate_custom_voice_from_text("hello world speak slowly", "speaker1", "english", options=opts, decode=False)
        b = m.generate_custom_voice_full(ids, "speaker1", "english", options=opts, decode=False)
        assert np.array_equal(a.codes, b.codes) and a.codes.shape[1] >= 1
        r = o.generate(np.asarray(ids, np.uint32), bind.resolve_codec_prefill(cfg, "english", "speaker1"), BASE, 6,
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"))
        assert np.array_equal(r["codes"], a.codes[0][:r["num_steps"]])
    finally:
        m.close()
