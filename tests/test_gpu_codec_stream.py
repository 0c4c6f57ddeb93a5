"""Streaming codec decode through the C ABI: the concatenation of the pushes must equal one whole decode of the same codes
(the reference has no streaming decode; its chunked decode is a different, inexact result and is tested elsewhere)."""
import numpy as np
import pytest

from conftest import snr_db

pytestmark = pytest.mark.gpu


def _stream(m, codes, pushes, max_chunk):
    out = []
    with m.open_codec_stream(codes.shape[0], max_chunk) as s:
        t = 0
        for n in pushes:
            out.append(s.push(codes[:, t:t + n]).copy())
            t += n
    assert t == codes.shape[1]
    return np.concatenate(out, axis=1), [o.shape[1] for o in out]


def test_stream_equals_whole_decode_f32(golden):
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True, talker=False)
    try:
        g = golden["codec_codes"]
        codes = np.stack([g[:300], g[300:600]])
        whole = m.decode_codes(codes)
        up = m.total_upsample()
        tail = up * 300 - len(whole[0])
        pushes = [5, 1, 7, 40, 3, 64, 64, 2, 50, 64]
        wav, sizes = _stream(m, codes, pushes, 64)
        assert sizes == [up * pushes[0] - tail] + [up * n for n in pushes[1:]]
        assert wav.shape[1] == len(whole[0])
        for b in range(2):
            assert np.abs(wav[b] - whole[b]).max() < 1e-5, b
            assert snr_db(whole[b], wav[b]) > 90.0
        # frame by frame (the smallest pushes) after a first push that decides at least one sample
        wav1, _ = _stream(m, codes[:, :40], [2] + [1] * 38, 2)
        assert np.abs(wav1[0] - whole[0][:wav1.shape[1]]).max() < 1e-5
        # and against the oracle's whole decode
        assert snr_db(golden["wav_full600_head"], wav[0][:4096]) >= 40.0
        # a session takes over the device buffers of the one closed before it on the handle: what they still hold (the other
        # row's keys, values and context) must not reach the samples
        wav2, _ = _stream(m, codes[::-1].copy(), pushes, 64)
        for b in range(2):
            assert np.abs(wav2[b] - whole[1 - b]).max() < 1e-5, b
    finally:
        m.close()


def test_stream_argument_errors(golden):
    from qwen3_tts_rs_amd import QttsError, synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True, talker=False)
    try:
        g = golden["codec_codes"]
        s = m.open_codec_stream(1, 8)
        with pytest.raises(QttsError):
            s.push(g[None, :9])                      # longer than max_chunk
        with pytest.raises(QttsError):
            s.push(np.stack([g[:4], g[4:8]]))        # batch mismatch
        bad = g[None, :4].copy()
        bad[0, 1, 0] = 1 << 20
        with pytest.raises(QttsError):
            s.push(bad)                              # code id out of range
        assert s.push(g[None, :8]).shape[1] > 0      # the session is still usable
        s.close()
        with pytest.raises(QttsError):
            s.push(g[None, :4])
        with pytest.raises(QttsError):
            m.open_codec_stream(0, 8)
    finally:
        m.close()


def test_stream_at_the_real_codec_shape_bf16():
    """Default codec shape (window 72, 1920 samples per frame), bf16: pushes of 25 frames, as a first-audio caller would."""
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("S06", dtype="bf16", codec=True, talker=False)
    try:
        codes = np.random.default_rng(5).integers(0, 2048, (2, 110, 16)).astype(np.uint32)
        whole = m.decode_codes(codes)
        wav, sizes = _stream(m, codes, [25, 25, 25, 25, 10], 25)
        assert sizes == [1920 * 25 - 555, 1920 * 25, 1920 * 25, 1920 * 25, 1920 * 10]
        for b in range(2):
            # a push and the whole decode pick different GEMM tilings (summation order), and every activation is stored in
            # bf16: two bf16 runs agree to the bf16 noise floor (~33 dB against f32), the f32 test above is the exact one
            assert snr_db(whole[b], wav[b]) > 36.0, b
    finally:
        m.close()
