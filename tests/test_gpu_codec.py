"""GPU parity of the codec decoder (codes -> waveform) through the C ABI.  Bar: waveform SNR >= 40 dB against the
oracle's decode of identical codes (fp32 runs reach > 100 dB; bf16 runs are bounded by bf16 activation storage)."""
import numpy as np
import pytest

from conftest import make_oracle, snr_db

pytestmark = pytest.mark.gpu

SNR_F32_DB = 40.0      # north-star bar
SNR_BF16_DB = 40.0     # the same bar: bf16 runs carry a remainder plane behind the activations that limit it


@pytest.fixture(scope="module")
def codec_f32():
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True, talker=False)
    yield m, ccfg
    m.close()


def test_full_decode_matches_golden(codec_f32, golden):
    m, ccfg = codec_f32
    codes = golden["codec_codes"]
    w = m.decode_codes(codes[None, :20])[0]
    assert len(w) == len(golden["wav_full20"])
    assert snr_db(golden["wav_full20"], w) >= SNR_F32_DB
    assert np.abs(w - golden["wav_full20"]).max() < 1e-4


def test_chunked_and_policy_decode_match_golden(codec_f32, golden):
    m, ccfg = codec_f32
    codes = golden["codec_codes"]
    w = m.decode_codes(codes[None, :33], chunk=8, left_ctx=3)[0]
    assert len(w) == len(golden["wav_chunked33"]) and snr_db(golden["wav_chunked33"], w) >= SNR_F32_DB
    w = m.decode_codes(codes[None], chunk=-2)[0]       # decode_output policy: 600 frames > 512 => 256 / 32
    assert len(w) == len(golden["wav_policy600"]) and snr_db(golden["wav_policy600"], w) >= SNR_F32_DB
    full = m.decode_codes(codes[None, :300])[0]
    assert snr_db(golden["wav_full600_head"], full[:4096]) >= SNR_F32_DB


def test_ragged_batch_and_edge_lengths(codec_f32, bind, golden):
    m, ccfg = codec_f32
    o, _, _, _ = make_oracle(bind, "T0", "f32", talker=False, codec=True)
    codes = golden["codec_codes"]
    lens = [20, 1, 0, 33, 2, 64, 65]
    T = max(lens)
    batch = np.zeros((len(lens), T, codes.shape[1]), np.uint32)
    for b, L in enumerate(lens):
        batch[b, :L] = codes[7 * b:7 * b + L]
    wavs = m.decode_codes(batch, lens)
    for b, L in enumerate(lens):
        if L == 0:
            assert len(wavs[b]) == 0
            continue
        ref = o.codec_decode(batch[b, :L])
        assert len(wavs[b]) == len(ref), f"row {b}"
        assert snr_db(ref, wavs[b]) >= SNR_F32_DB, f"row {b}"


def test_bad_codes_are_rejected(codec_f32, golden):
    from qwen3_tts_rs_amd import QttsError
    m, ccfg = codec_f32
    bad = golden["codec_codes"][None, :8].copy()
    bad[0, 3, 1] = ccfg["decoder_config"]["codebook_size"]
    with pytest.raises(QttsError):
        m.decode_codes(bad)


def test_bf16_decode(bind, golden):
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="bf16", codec=True, talker=False, round_to_bf16=True)
    o, _, _, _ = make_oracle(bind, "T0", "bf16", talker=False, codec=True)
    try:
        codes = golden["codec_codes"][:40]
        w = m.decode_codes(codes[None])[0]
        ref = o.codec_decode(codes)
        assert len(w) == len(ref)
        assert snr_db(ref, w) >= SNR_BF16_DB
    finally:
        m.close()


@pytest.mark.parametrize("dtype,causal_db", [("bf16", 36.0), ("f32", 80.0)])
def test_full_size_codec_properties(dtype, causal_db):
    """Reference-default codec shapes (1024-wide transformer, 1536-channel vocoder, x1920 upsampling) through properties that
    need no oracle (the oracle comparison at these shapes, rows of up to 104 frames, is in test_gpu_parity_full_size.py):
    causality (a prefix decodes to a prefix; in bf16 only down to the rounding
    noise floor because the row count selects a different GEMM summation order), batch-row independence, chunked
    decode equals full decode on the first window, samples clamped to [-1, 1]."""
    import torch
    from qwen3_tts_rs_amd import Model, LoaderConfig, synth
    from conftest import SEED
    cfg, ccfg, tmpl = synth.shape_set("S06")
    m = Model(cfg, ccfg, LoaderConfig(dtype=dtype))
    for k, t in synth.iter_weights_torch(synth.codec_keys(ccfg), SEED, "cuda:0", dtype == "bf16"):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    try:
        rng = np.random.default_rng(1)
        codes = rng.integers(0, 2048, (3, 140, 16)).astype(np.uint32)
        codes[2] = codes[0]
        wavs = m.decode_codes(codes, [140, 90, 140])
        up = 1920
        assert [len(w) for w in wavs] == [140 * up - 555, 90 * up - 555, 140 * up - 555]
        assert all(np.isfinite(w).all() and np.abs(w).max() <= 1.0 for w in wavs)
        assert np.array_equal(wavs[0], wavs[2])                       # rows are independent of their batch position
        prefix = m.decode_codes(codes[:1, :100])[0]
        n = len(prefix) - 2 * up                                     # the vocoder's trimmed tail differs, the rest is causal
        assert snr_db(prefix[:n], wavs[0][:n]) > causal_db
        ch = m.decode_codes(codes[:1], chunk=100, left_ctx=25)[0]
        assert len(ch) == 100 * up - 555 + 40 * up - 555
        assert snr_db(m.decode_codes(codes[:1, :100])[0], ch[:100 * up - 555]) > causal_db
    finally:
        m.close()


def test_decode_matches_the_independent_code2wav_golden(codec_f32):
    """The GPU decoder against waveforms from transformers' Code2Wav modules on the same weights (tests/golden/
    make_code2wav_golden.py) - the same fixture that pins the CPU oracle, with no oracle in between."""
    import os
    m, ccfg = codec_f32
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "code2wav_golden.npz"))
    for name in ("a", "b"):
        w = m.decode_codes(g["codes_" + name][None])[0]
        ref = g["wav_" + name]
        assert len(w) == len(ref)
        assert snr_db(ref, w) >= 80.0 and np.abs(w - ref).max() < 1e-4, name
