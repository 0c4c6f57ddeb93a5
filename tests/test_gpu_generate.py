"""GPU parity of the generate path (prompt -> talker -> code predictor -> sampling) through the C ABI.
Bar: fp32 greedy / seeded token ids bit-exact against the CPU oracle; bf16 logits within max-abs 2e-2."""
import numpy as np
import pytest

from conftest import make_oracle
from golden.make_golden import GREEDY, SAMPLED

pytestmark = pytest.mark.gpu

BF16_LOGIT_TOL = 2e-2


def _sampling(d):
    from qwen3_tts_rs_amd import SamplingConfig
    return SamplingConfig(**{k: d[k] for k in SamplingConfig().__dict__ if k in d})


@pytest.fixture(scope="module")
def t0_f32():
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=False)
    yield m, cfg, tmpl
    m.close()


def _req(text, **kw):
    from qwen3_tts_rs_amd import GenerationRequest
    return GenerationRequest(text_ids=text, language=kw.pop("language", "english"), speaker=kw.pop("speaker", "speaker1"), **kw)


def test_greedy_matches_golden_tokens(t0_f32, golden):
    m, cfg, tmpl = t0_f32
    out = m.generate([_req(golden["text_ids"])], _sampling(GREEDY), 64, trace=True, use_graph=False)
    assert out.num_steps == 64 and out.effective_lengths == [64]
    assert np.array_equal(out.codes[0], golden["greedy_codes"])
    assert np.abs(out.logits[0, :4] - golden["greedy_logits_head"]).max() < 1e-4
    assert np.abs(out.hidden[0, :4] - golden["greedy_hidden_head"]).max() < 1e-4
    out2 = m.generate([_req(golden["text_ids"])], _sampling(GREEDY), 64, use_graph=True)
    assert np.array_equal(out2.codes, out.codes)     # hipGraph replay == eager launches


def test_seeded_sampling_matches_golden_tokens(t0_f32, golden):
    m, cfg, tmpl = t0_f32
    out = m.generate([_req(golden["text_ids"], seed=42)], _sampling(SAMPLED), 64)
    assert np.array_equal(out.codes[0], golden["sampled_codes"])


def test_eos_handling(golden):
    from qwen3_tts_rs_amd import Model, LoaderConfig, synth
    from conftest import SEED
    cfg, ccfg, tmpl = synth.shape_set("T0")
    cfg["talker_config"]["codec_eos_token_id"] = int(golden["eos_id"])
    m = Model(cfg, None, LoaderConfig(dtype="f32"))
    m.load_weights(synth.iter_weights(synth.talker_keys(cfg), SEED))
    m.finalize()
    try:
        # second row never reaches EOS within the budget of the first: the loop ends when every row has hit EOS or
        # at max_new_tokens, rows keep generating after their EOS (nn/generation.rs:996-1000,1079)
        out = m.generate([_req(golden["text_ids"])], _sampling(GREEDY), 64)
        assert out.num_steps == int(golden["eos_num_steps"])
        assert out.effective_lengths == [int(golden["eos_effective_len"])]
        assert np.array_equal(out.codes[0], golden["eos_codes"])
        # fixed-length measurement protocol: EOS stays suppressed for min_new_tokens frames, the frames before the
        # reference's EOS step are unchanged and the run goes on to the requested length
        k = int(golden["eos_effective_len"])
        fixed = m.generate([_req(golden["text_ids"])], _sampling(GREEDY), 64, min_new_tokens=64)
        assert fixed.num_steps == 64 and fixed.effective_lengths == [64]
        assert np.array_equal(fixed.codes[0][:k], golden["eos_codes"][:k])
        assert (fixed.codes[0][:, 0] != int(golden["eos_id"])).all()
    finally:
        m.close()


def test_ragged_batch_is_token_identical_per_utterance(t0_f32, bind):
    m, cfg, tmpl = t0_f32
    from qwen3_tts_rs_amd import synth
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    texts = [synth.synthetic_text_ids(tmpl, n, i) for i, n in enumerate([10, 31, 1, 17, 24, 12, 9])]
    langs = ["english", "chinese", "auto", "english", "auto", "chinese", "english"]
    spks = ["speaker0", "speaker1", "speaker2", "speaker1", "speaker0", "speaker2", "speaker1"]
    reqs = [_req(t, language=l, speaker=s) for t, l, s in zip(texts, langs, spks)]
    out = m.generate(reqs, _sampling(GREEDY), 16, trace=True)
    for b, (t, l, s) in enumerate(zip(texts, langs, spks)):
        r = o.generate(t, bind.resolve_codec_prefill(cfg, l, s), GREEDY, 16, speaker_token=bind.resolve_speaker_token(cfg, s), trace=True)
        assert np.array_equal(r["codes"], out.codes[b]), f"row {b}"
        assert np.abs(r["logits"] - out.logits[b]).max() < 1e-4


@pytest.mark.parametrize("batch", [130, 300])
def test_wide_batches_stay_token_identical(t0_f32, bind, batch):
    """More than 128 rows per decode step: the split-K GEMM runs one block per 64-row slice (up to 512 rows).  Rows spread over
    the slices are compared with the oracle run one utterance at a time."""
    m, cfg, tmpl = t0_f32
    from qwen3_tts_rs_amd import synth
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    texts = [synth.synthetic_text_ids(tmpl, 3 + (i * 7) % 23, i) for i in range(batch)]
    out = m.generate([_req(t, speaker=f"speaker{i % 3}") for i, t in enumerate(texts)], _sampling(GREEDY), 6)
    for b in (0, 63, 64, 127, 128, batch - 1):
        s = f"speaker{b % 3}"
        r = o.generate(texts[b], bind.resolve_codec_prefill(cfg, "english", s), GREEDY, 6, speaker_token=bind.resolve_speaker_token(cfg, s))
        assert np.array_equal(r["codes"], out.codes[b]), f"row {b}"


def test_voice_design_and_xvector_prompts(t0_f32, bind, golden):
    m, cfg, tmpl = t0_f32
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    from qwen3_tts_rs_amd import GenerationRequest
    reqs = [GenerationRequest(text_ids=golden["text_ids"], language="auto", instruct_ids=golden["instruct_ids"]),
            GenerationRequest(text_ids=golden["text_ids"], language="english", speaker_embed=golden["xvector"])]
    out = m.generate(reqs, _sampling(GREEDY), 12)
    r0 = o.generate(golden["text_ids"], bind.resolve_codec_prefill(cfg, "auto", None), GREEDY, 12, instruct_ids=golden["instruct_ids"])
    r1 = o.generate(golden["text_ids"], bind.resolve_codec_prefill(cfg, "english", None), GREEDY, 12, speaker_embed=golden["xvector"])
    assert np.array_equal(out.codes[0], r0["codes"]) and np.array_equal(out.codes[1], r1["codes"])


@pytest.mark.parametrize("non_streaming", [False, True])
def test_icl_prompt_matches_golden(t0_f32, bind, golden, non_streaming):
    """Voice clone with in-context reference codes + transcript (generate_icl_prompt, nn/generation.rs:292-416)."""
    from qwen3_tts_rs_amd import GenerationRequest
    m, cfg, tmpl = t0_f32
    req = GenerationRequest(text_ids=golden["text_ids"], language="english", speaker_embed=golden["xvector"],
                            ref_codes=golden["ref_codes"], ref_text_ids=golden["ref_text_ids"], non_streaming_mode=non_streaming)
    out = m.generate([req, _req(golden["text_ids"])], _sampling(GREEDY), 24)
    if not non_streaming:
        assert np.array_equal(out.codes[0], golden["icl_codes"])
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    r = o.generate(golden["text_ids"], bind.resolve_codec_prefill(cfg, "english", None), GREEDY, 24, speaker_embed=golden["xvector"],
                   ref_text_ids=golden["ref_text_ids"], ref_codes=golden["ref_codes"], non_streaming=non_streaming)
    assert np.array_equal(out.codes[0], r["codes"])
    assert np.array_equal(out.codes[1], golden["greedy_codes"][:24])      # the other row of the batch is unaffected
    # a short text makes the reference codec side longer than the text side (third layout of generate_icl_prompt)
    short = golden["text_ids"][[0, 1, 2, 3, -5, -4, -3, -2, -1]]
    req2 = GenerationRequest(text_ids=short, language="english", speaker_embed=golden["xvector"], ref_codes=golden["ref_codes"],
                             ref_text_ids=golden["ref_text_ids"][[0, 1, 2, 3, -2, -1]], non_streaming_mode=non_streaming)
    out2 = m.generate([req2], _sampling(GREEDY), 8)
    r2 = o.generate(short, bind.resolve_codec_prefill(cfg, "english", None), GREEDY, 8, speaker_embed=golden["xvector"],
                    ref_text_ids=golden["ref_text_ids"][[0, 1, 2, 3, -2, -1]], ref_codes=golden["ref_codes"], non_streaming=non_streaming)
    assert np.array_equal(out2.codes[0], r2["codes"])


def test_first_audio_chunk_is_a_final_prefix(golden):
    """The first-audio hook hands over the first frames while generation goes on; decoding that prefix gives the same
    samples as the full decode for all but its last two frames (the codec decoder is causal)."""
    from conftest import snr_db
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True)
    try:
        got = {}

        def on_first(codes):
            got["codes"] = codes
            got["wav"] = m.decode_codes(codes)

        reqs = [_req(golden["text_ids"]), _req(golden["text_ids"][:15].tolist() + golden["text_ids"][-5:].tolist())]
        out = m.generate(reqs, _sampling(GREEDY), 40, first_frames=12, on_first_chunk=on_first)
        assert np.array_equal(out.codes[0], golden["greedy_codes"][:40])
        assert got["codes"].shape == (2, 12, 4) and np.array_equal(got["codes"], out.codes[:, :12])
        full = m.decode_output(out)
        up = 96
        for b in range(2):
            n = up * (12 - 2)
            assert snr_db(full[b][:n], got["wav"][b][:n]) > 60.0
        # a generation shorter than the chunk still delivers it once, at the end
        got.clear()
        out = m.generate(reqs[:1], _sampling(GREEDY), 5, first_frames=12, on_first_chunk=on_first)
        assert got["codes"].shape == (1, 5, 4) and np.array_equal(got["codes"], out.codes[:, :5])
    finally:
        m.close()


def test_streaming_text_mode(t0_f32, bind, golden):
    """non_streaming_mode = false: the text after the first token is fed one row per generated frame
    (nn/generation.rs:769-789,1096-1106)."""
    m, cfg, tmpl = t0_f32
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=True, codec=False)
    out = m.generate([_req(golden["text_ids"], non_streaming_mode=False)], _sampling(GREEDY), 20)
    r = o.generate(golden["text_ids"], bind.resolve_codec_prefill(cfg, "english", "speaker1"), GREEDY, 20,
                   speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), non_streaming=False)
    assert np.array_equal(out.codes[0], r["codes"])


def test_argument_errors(t0_f32, golden):
    from qwen3_tts_rs_amd import QttsError
    m, cfg, tmpl = t0_f32
    with pytest.raises(QttsError):
        m.generate([_req(golden["text_ids"][:3])], _sampling(GREEDY), 4)         # too short for the chat template
    # an unknown speaker name is dropped silently by build_prompt (nn/generation.rs:551-566); the validation error
    # belongs to the Model-level API (model/mod.rs:341-419)
    a = m.generate([_req(golden["text_ids"], speaker="nobody")], _sampling(GREEDY), 4)
    b = m.generate([_req(golden["text_ids"], speaker=None)], _sampling(GREEDY), 4)
    assert np.array_equal(a.codes, b.codes)
    with pytest.raises(QttsError, match="Unknown speaker"):
        m.generate_custom_voice_full(golden["text_ids"], "nobody", "english", decode=False)
    with pytest.raises(QttsError, match="Unsupported language"):
        m.generate_custom_voice_full(golden["text_ids"], "speaker1", "klingon", decode=False)
    with pytest.raises(QttsError):
        m.generate([_req(np.array([99999] * 12, np.uint32))], _sampling(GREEDY), 4)  # id outside the text table


@pytest.mark.parametrize("shape,B,steps", [("M0", 20, 8)])
def test_mid_size_f32_tokens_exact(bind, shape, B, steps):
    """Realistic tile shapes (head_dim 128, 16 codebooks, vocab 3072) and a batch that exercises the multi-tile GEMMs."""
    from qwen3_tts_rs_amd import synth, synthetic_model
    m, cfg, _, tmpl = synthetic_model(shape, dtype="f32", codec=False)
    o, _, _, _ = make_oracle(bind, shape, "f32", talker=True, codec=False)
    try:
        texts = [synth.synthetic_text_ids(tmpl, 8 + (5 * b) % 23, b) for b in range(B)]
        out = m.generate([_req(t) for t in texts], _sampling(GREEDY), steps, trace=True)
        for b in (0, 7, B - 1):
            r = o.generate(texts[b], bind.resolve_codec_prefill(cfg, "english", "speaker1"), GREEDY, steps,
                           speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), trace=True)
            assert np.array_equal(r["codes"], out.codes[b])
            assert np.abs(r["logits"] - out.logits[b]).max() < 1e-4
    finally:
        m.close()


@pytest.mark.parametrize("shape", ["T0", "M0"])
def test_bf16_logits_within_tolerance(bind, shape):
    """bf16 weights / KV / GEMM inputs with f32 accumulation against the f32 oracle on the same bf16-rounded weights.
    Logits are compared frame by frame while the two token histories still agree (after a near-tie flips a token
    the inputs differ and the comparison is meaningless)."""
    from qwen3_tts_rs_amd import synth, synthetic_model
    m, cfg, _, tmpl = synthetic_model(shape, dtype="bf16", codec=False, round_to_bf16=True)
    o, _, _, _ = make_oracle(bind, shape, "bf16", talker=True, codec=False)
    try:
        text = synth.synthetic_text_ids(tmpl, 14, 3)
        out = m.generate([_req(text)], _sampling(GREEDY), 8, trace=True)
        r = o.generate(text, bind.resolve_codec_prefill(cfg, "english", "speaker1"), GREEDY, 8,
                       speaker_token=bind.resolve_speaker_token(cfg, "speaker1"), trace=True)
        agree = 0
        while agree < 8 and np.array_equal(r["codes"][agree], out.codes[0, agree]):
            agree += 1
        n = min(8, agree + 1)     # the first diverging frame still has identical inputs to the talker
        assert n >= 1
        assert np.abs(r["logits"][:n] - out.logits[0, :n]).max() <= BF16_LOGIT_TOL
    finally:
        m.close()


def test_full_size_properties():
    """1.7B-like shape set, batch 64, bf16: sizes the oracle cannot reach in seconds are covered by properties of the
    algorithm: determinism, hipGraph == eager, code ranges, suppressed ids never sampled, batch-row independence of
    the prompt (identical requests give identical rows)."""
    import torch
    from qwen3_tts_rs_amd import Model, LoaderConfig, synth
    from conftest import SEED
    cfg, ccfg, tmpl = synth.shape_set("S17")
    m = Model(cfg, None, LoaderConfig(dtype="bf16"))
    for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), SEED, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    try:
        B, steps = 64, 6
        texts = [synth.synthetic_text_ids(tmpl, 8 + (7 * b) % 40, b // 2) for b in range(B)]
        texts[1] = texts[0]
        reqs = [_req(t) for t in texts]
        a = m.generate(reqs, _sampling(GREEDY), steps, use_graph=True)
        b = m.generate(reqs, _sampling(GREEDY), steps, use_graph=False)
        assert np.array_equal(a.codes, b.codes)
        assert np.array_equal(a.codes[0], a.codes[1])
        t = cfg["talker_config"]
        assert (a.codes[:, :, 0] < t["vocab_size"] - 1024).all() or (a.codes[:, :, 0] == t["codec_eos_token_id"]).any()
        assert (a.codes[:, :, 1:] < t["code_predictor_config"]["vocab_size"]).all()
        assert len(np.unique(a.codes[:, 0, 0])) > 1 or B == 1
    finally:
        m.close()
