"""GPU parity against the CPU oracle AT THE BENCHMARKED SHAPES (0.6B-like and 1.7B-like talkers, the reference-default
codec), through the C ABI.  The oracle gets exactly the tensors the GPU model gets (generated once on the device,
copied to the host), runs one utterance at a time with its OpenMP build and finishes a few frames of these shapes in
seconds.

Bars: f32 greedy code ids bit-exact (a first difference is accepted only where the oracle's own top-2 logit margin at that
step is below F32_TIE_MARGIN, and is reported); f32 logits within 1e-3; bf16 logits within max-abs 2e-2 for as long as
the code histories agree; waveform SNR >= 40 dB."""
import numpy as np
import pytest

from conftest import SEED, snr_db
from golden.make_golden import GREEDY, SAMPLED

pytestmark = pytest.mark.gpu

F32_LOGIT_TOL = 1e-3
F32_TIE_MARGIN = 1e-4
BF16_LOGIT_TOL = 2e-2
SNR_DB = 40.0


def _greedy():
    from qwen3_tts_rs_amd import SamplingConfig
    return SamplingConfig(**{k: GREEDY[k] for k in SamplingConfig().__dict__ if k in GREEDY})


def _pair(bind, shape, dtype, talker=True, codec=False):
    """The GPU model and the oracle loaded with the same tensors (bf16 runs: both hold the bf16-rounded values)."""
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set(shape, "custom_voice")
    m = Model(cfg, ccfg if codec else None, LoaderConfig(dtype=dtype))
    o = bind.Oracle(fast=True)
    keys = (synth.talker_keys(cfg) if talker else []) + (synth.codec_keys(ccfg) if codec else [])
    for k, t in synth.iter_weights_torch(keys, SEED, "cuda:0", dtype == "bf16"):
        if dtype == "bf16" and t.dtype == torch.float32:
            t = t.to(torch.bfloat16).float()       # vectors / codebooks: rounded too, whatever each side stores them as
        m.load_tensor(k, t)
        o.load_tensor(k, t.float().cpu().numpy())
        del t
    m.finalize()
    torch.cuda.empty_cache()
    if talker:
        o.set_model_config(cfg)
    if codec:
        o.set_codec_config(ccfg)
    return m, o, cfg, ccfg, tmpl


def _oracle_row(bind, o, cfg, text, frames, speaker="speaker1", sampling=GREEDY, seed=0):
    return o.generate(text, bind.resolve_codec_prefill(cfg, "english", speaker), sampling, frames, seed=seed,
                      speaker_token=bind.resolve_speaker_token(cfg, speaker), trace=True)


def _check_f32_row(ref, codes, logits, label):
    """Token identity with the margin rule, logits while the histories agree."""
    n = ref["num_steps"]
    assert codes.shape[0] == n, label
    same = (ref["codes"] == codes).all(axis=1)
    first_bad = int(np.argmin(same)) if not same.all() else n
    upto = min(first_bad + 1, n)             # the logits of the first differing frame still come from equal histories
    assert np.abs(ref["logits"][:upto] - logits[:upto]).max() < F32_LOGIT_TOL, label
    if first_bad < n:
        top2 = np.sort(ref["logits"][first_bad])[-2:]
        margin = float(top2[1] - top2[0])
        cb0_differs = ref["codes"][first_bad, 0] != codes[first_bad, 0]
        assert cb0_differs and margin < F32_TIE_MARGIN, (
            f"{label}: codes differ at frame {first_bad} (oracle top-2 margin {margin:.3g}, minimum margin of the run "
            f"{ref['min_margin']:.3g})")
        print(f"{label}: near-tie at frame {first_bad}, oracle margin {margin:.3g}")


def test_small_model_custom_voice_greedy_is_token_identical():
    """0.6B-like CustomVoice model, one short sentence (12 content tokens), CLI-greedy, f32: the plumbing configuration,
    the GPU run against the CPU run of the same inputs and weights."""
    from oracle import bind
    from qwen3_tts_rs_amd import GenerationRequest, synth
    m, o, cfg, ccfg, tmpl = _pair(bind, "S06", "f32")
    try:
        text = synth.synthetic_text_ids(tmpl, 12, 0)
        frames = 20
        ref = _oracle_row(bind, o, cfg, text, frames)
        out = m.generate([GenerationRequest(text_ids=text, language="english", speaker="speaker1")], _greedy(), frames, trace=True)
        _check_f32_row(ref, out.codes[0], out.logits[0], "S06 f32 B=1")
        graph = m.generate([GenerationRequest(text_ids=text, language="english", speaker="speaker1")], _greedy(), frames)
        assert np.array_equal(graph.codes, out.codes)
    finally:
        m.close()


def _batch64(tmpl):
    from qwen3_tts_rs_amd import GenerationRequest, synth
    texts = [synth.synthetic_text_ids(tmpl, 8 + (u * 37 + 11) % 57, u) for u in range(64)]
    return texts, [GenerationRequest(text_ids=t, language="english", speaker="speaker1") for t in texts]


def test_headline_shape_f32_rows_of_a_batch_of_64():
    """1.7B-like model, f32, the benchmark's 64 texts as one batch: rows 0, 31 and 63 against the oracle run one utterance
    at a time - code ids exact, logits within 1e-3."""
    from oracle import bind
    m, o, cfg, ccfg, tmpl = _pair(bind, "S17", "f32")
    try:
        frames = 8
        texts, reqs = _batch64(tmpl)
        out = m.generate(reqs, _greedy(), frames, trace=True)
        graph = m.generate(reqs, _greedy(), frames)
        assert np.array_equal(graph.codes, out.codes)          # hipGraph replay == kernel-by-kernel launches
        for b in (0, 31, 63):
            ref = _oracle_row(bind, o, cfg, texts[b], frames)
            _check_f32_row(ref, out.codes[b], out.logits[b], f"S17 f32 row {b}")
    finally:
        m.close()


def test_headline_shape_f32_rows_at_the_benchmarked_context_lengths():
    """1.7B-like model, f32, the benchmark's batch of 64 for 100 frames: prompts of 19 .. 75 positions grow past the 64-token
    KV page twice (a row's cache then spans three pages).  Rows 0, 31 and 63 against the oracle run one utterance at a time:
    code ids exact under the margin rule, logits within 1e-3 for as long as the histories agree."""
    from oracle import bind
    m, o, cfg, ccfg, tmpl = _pair(bind, "S17", "f32")
    try:
        frames = 100
        texts, reqs = _batch64(tmpl)
        out = m.generate(reqs, _greedy(), frames, trace=True)
        graph = m.generate(reqs, _greedy(), frames)
        assert np.array_equal(graph.codes, out.codes)
        for b in (0, 31, 63):
            ref = _oracle_row(bind, o, cfg, texts[b], frames)
            assert ref["num_steps"] == frames
            _check_f32_row(ref, out.codes[b], out.logits[b], f"S17 f32 row {b}, {frames} frames")
    finally:
        m.close()


def test_headline_shape_f32_seeded_sampling_as_the_benchmark_draws():
    """The benchmark's sampling mode at its shape: temperature 0.9, top-k 50 for the talker and the code predictor, one
    MT19937 stream per utterance seeded 42 + u.  f32, rows 0, 31 and 63 of the batch of 64, 20 frames: every code id equals the
    oracle's (the draw compares a 24-bit uniform with f32 running sums, so a difference needs a cumulative probability
    within rounding of the uniform; none occurs on these rows)."""
    from oracle import bind
    from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig
    m, o, cfg, ccfg, tmpl = _pair(bind, "S17", "f32")
    try:
        frames = 20
        texts, _ = _batch64(tmpl)
        reqs = [GenerationRequest(text_ids=t, language="english", speaker="speaker1", seed=42 + u) for u, t in enumerate(texts)]
        sc = SamplingConfig(**{k: SAMPLED[k] for k in SamplingConfig().__dict__ if k in SAMPLED})
        out = m.generate(reqs, sc, frames)
        eager = m.generate(reqs, sc, frames, use_graph=False)
        assert np.array_equal(eager.codes, out.codes)
        for b in (0, 31, 63):
            ref = _oracle_row(bind, o, cfg, texts[b], frames, sampling=SAMPLED, seed=42 + b)
            n = ref["num_steps"]
            assert n == out.codes[b].shape[0] or n <= frames
            assert np.array_equal(ref["codes"][:n], out.codes[b][:n]), (
                f"row {b}: first difference at frame {int(np.argmin((ref['codes'][:n] == out.codes[b][:n]).all(axis=1)))}")
    finally:
        m.close()


def test_headline_shape_bf16_logits_beyond_the_first_kv_page():
    """bf16 decode attention over more than one KV page at the benchmarked shape: the longest prompts of the batch of 64 (75
    positions) start their first frame on the second page.  Logits of those rows within 2e-2 of the oracle for as long as the
    code histories agree."""
    from oracle import bind
    m, o, cfg, ccfg, tmpl = _pair(bind, "S17", "bf16")
    try:
        frames = 6
        texts, reqs = _batch64(tmpl)
        longest = sorted(range(64), key=lambda u: -len(texts[u]))[:2]
        assert all(len(texts[u]) - 8 + 3 + 5 + 2 > 64 for u in longest)      # prompt positions: role + codec prefill + text + eos + bos
        out = m.generate(reqs, _greedy(), frames, trace=True)
        for b in longest:
            ref = _oracle_row(bind, o, cfg, texts[b], frames)
            same = (ref["codes"] == out.codes[b]).all(axis=1)
            upto = min((int(np.argmin(same)) if not same.all() else frames) + 1, frames)
            err = np.abs(ref["logits"][:upto] - out.logits[b][:upto]).max()
            assert err <= BF16_LOGIT_TOL, f"row {b}: {err:.4f} over {upto} frames"
    finally:
        m.close()


def _gpu_model(shape, dtype):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set(shape, "custom_voice")
    m = Model(cfg, None, LoaderConfig(dtype=dtype))
    for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), SEED, "cuda:0", dtype == "bf16"):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    return m, tmpl


def test_kernel_choice_is_fixed_per_weight_set():
    """The decode kernels of a bf16 model are picked once per weight set - for one stream when its first generate call comes
    before any session, for several streams when a session is opened first - and the two choices sum in different orders.  A
    request's codes must therefore not depend on whether a session was opened later, on any model."""
    m, tmpl = _gpu_model("S17", "bf16")
    other = None
    try:
        frames = 6
        _, reqs = _batch64(tmpl)
        first = m.generate(reqs, _greedy(), frames)
        ses = m.open_session()                       # opened after the first call: the weight set keeps its choice
        other, _ = _gpu_model("S17", "bf16")
        ses_other = other.open_session()             # a second weight set, sessions first: the other choice, its own business
        assert np.array_equal(m.generate(reqs, _greedy(), frames).codes, first.codes)
        assert np.array_equal(ses.generate(reqs, _greedy(), frames).codes, first.codes)
        a = other.generate(reqs, _greedy(), frames)
        b = ses_other.generate(reqs, _greedy(), frames)
        assert np.array_equal(a.codes, b.codes)
        assert np.array_equal(m.generate(reqs, _greedy(), frames).codes, first.codes)
        ses.close()
        ses_other.close()
    finally:
        m.close()
        if other is not None:
            other.close()


@pytest.mark.parametrize("with_session", [False, True], ids=["one-stream-kernels", "several-streams-kernels"])
def test_headline_shape_bf16_logits_of_a_batch_of_64(with_session):
    """The benchmarked configuration itself (1.7B-like, bf16, 64 rows): logits of rows 0, 31 and 63 within max-abs 2e-2 of
    the oracle on the same bf16-rounded weights for as long as the code histories agree (at least the first frame).
    With a session open the decode kernels are the ones picked for several streams in flight (the benchmark's case): the run
    then goes through the session's handle."""
    from oracle import bind
    base, o, cfg, ccfg, tmpl = _pair(bind, "S17", "bf16")
    m = base.open_session() if with_session else base
    try:
        frames = 8
        texts, reqs = _batch64(tmpl)
        out = m.generate(reqs, _greedy(), frames, trace=True)
        alone = {b: m.generate([reqs[b]], _greedy(), frames, trace=True) for b in (0, 31, 63)}
        if with_session:   # the captured frame graph of the same kernels reproduces the kernel-by-kernel launches
            assert np.array_equal(m.generate(reqs, _greedy(), frames).codes, out.codes)
        for b in (0, 31, 63):
            ref = _oracle_row(bind, o, cfg, texts[b], frames)
            same = (ref["codes"] == out.codes[b]).all(axis=1)
            upto = min((int(np.argmin(same)) if not same.all() else frames) + 1, frames)
            err = np.abs(ref["logits"][:upto] - out.logits[b][:upto]).max()
            assert err <= BF16_LOGIT_TOL, f"row {b}: {err:.4f} over {upto} frames"
            # the same utterance alone and inside the batch: bounded by the same tolerance while histories agree
            same_a = (alone[b].codes[0] == out.codes[b]).all(axis=1)
            upto_a = min((int(np.argmin(same_a)) if not same_a.all() else frames) + 1, frames)
            assert np.abs(alone[b].logits[0][:upto_a] - out.logits[b][:upto_a]).max() <= BF16_LOGIT_TOL, f"row {b} alone vs batch"
            # the code predictor's logits, given the hidden state and first code the GPU run itself produced: every codebook
            # of a frame up to (and including) the first one where the two pick different codes
            for t in range(upto):
                oc, ol = o.code_predictor(out.hidden[b][t], int(out.codes[b][t][0]), GREEDY)
                agree = oc == out.codes[b][t][1:]
                nj = min((int(np.argmin(agree)) if not agree.all() else len(oc)) + 1, len(oc))
                cerr = np.abs(ol[:nj] - out.cp_logits[b][t][:nj]).max()
                assert cerr <= BF16_LOGIT_TOL, f"row {b} frame {t}: code-predictor logits {cerr:.4f} over {nj} codebooks"
    finally:
        if with_session:
            m.close()
        base.close()


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reference_default_codec_against_the_oracle(dtype):
    """Reference-default codec shapes (1024-wide transformer, 1536-channel vocoder, x1920): rows of 104, 12, 9, 1, 2 and 0
    frames against the oracle's decode of identical codes.  The long row crosses the 72-key attention window and spans two
    64-query tiles of the banded attention kernel; the rows of one and two frames are shorter than a row tile of the fused
    vocoder units at every rate (their first tile is also their last), the empty row yields no samples."""
    from oracle import bind
    m, o, cfg, ccfg, tmpl = _pair(bind, "S17", dtype, talker=False, codec=True)
    try:
        rng = np.random.default_rng(5)
        codes = rng.integers(0, 2048, (6, 104, 16)).astype(np.uint32)
        lens = [104, 12, 9, 1, 2, 0]
        wavs = m.decode_codes(codes, lens)
        for b, L in enumerate(lens):
            if L == 0:
                assert len(wavs[b]) == 0
                continue
            ref = o.codec_decode(codes[b, :L])
            assert len(ref) == len(wavs[b]) == 1920 * L - 555
            s = snr_db(ref, wavs[b])
            assert s >= SNR_DB, f"{dtype} row {b}: {s:.1f} dB"
    finally:
        m.close()


def test_long_form_codec_decode_batch_32():
    """Codec standalone at the configuration's own size: 32 sequences of 7500 frames (10 minutes) in one call, chunked by the
    decode_output policy (256 / 32).  Size-independent properties: sample counts, range, rows with equal codes give equal
    waveforms wherever they sit in the batch, and a window re-decoded alone matches."""
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set("S17", "custom_voice")
    m = Model(cfg, ccfg, LoaderConfig(dtype="bf16"))
    for k, t in synth.iter_weights_torch(synth.codec_keys(ccfg), SEED, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    try:
        rng = np.random.default_rng(3)
        B, T = 32, 7500
        codes = rng.integers(0, 2048, (B, T, 16)).astype(np.uint32)
        codes[17] = codes[2]
        wavs = m.decode_codes(codes, [T] * B, chunk=-2)
        nchunks = -(-T // 256)
        for w in wavs:
            assert len(w) == 1920 * T - 555 * nchunks
        assert all(np.isfinite(w).all() and np.abs(w).max() <= 1.0 for w in wavs)
        assert np.array_equal(wavs[2], wavs[17])
        part = m.decode_codes(codes[5:6, 224:512], chunk=0)[0]          # second window of row 5 with its 32 context frames
        ref = wavs[5][1920 * 256 - 555:][:1920 * 256 - 555]
        n = min(len(ref), len(part) - 32 * 1920)
        assert snr_db(part[32 * 1920:32 * 1920 + n], ref[:n]) > 36.0
    finally:
        m.close()
