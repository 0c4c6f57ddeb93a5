"""The CPU oracle against every known-answer vector the reference's own unit tests hold for this path
(tests/golden/reference_kats.json cites each source)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN


@pytest.fixture(scope="module")
def kats():
    with open(os.path.join(GOLDEN, "reference_kats.json")) as f:
        return json.load(f)


def _vals(xs):
    return np.array([-np.inf if x == "-inf" else float(x) for x in xs], np.float32)


def _proc(lib, which, logits, arg=0.0, hist=()):
    l = np.array(logits, np.float32)
    h = np.array(list(hist), np.int64)
    lib.orc_processor(which, l.ctypes.data_as(C.c_void_p), len(l), C.c_double(arg), h.ctypes.data_as(C.c_void_p), len(h))
    return l


def _sample(bind, logits, hist=(), **cfg):
    lib = bind.lib()
    l = np.array(logits, np.float32)
    h = np.array(list(hist), np.int64)
    s = bind.sampling_array(dict(do_sample=False, temperature=1.0, top_k=0, repetition_penalty=1.0, min_new_tokens=0, **cfg))
    return lib.orc_sample_token(l.ctypes.data_as(C.c_void_p), len(l), s.ctypes.data_as(C.c_void_p), h.ctypes.data_as(C.c_void_p),
                                len(h), 0)


def test_randn_vectorized_seed42(bind, kats):
    k = kats["randn_vectorized_seed42_first8"]
    out = np.zeros(k["count"], np.float32)
    bind.lib().orc_randn_vectorized(k["seed"], k["count"], out.ctypes.data_as(C.c_void_p))
    assert np.abs(out[:8] - np.array(k["expected"], np.float32)).max() < k["tol"]
    assert np.isfinite(out).all()


def test_mt19937_reference_stream(bind):
    # MT19937 init_genrand(5489) first outputs (Matsumoto & Nishimura's published reference stream)
    out = np.zeros(3, np.uint32)
    bind.lib().orc_mt_u32(5489, 3, out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == [3499211612, 581869302, 3890346734]
    u = np.zeros(100, np.float32)
    bind.lib().orc_mt_uniform(42, 100, u.ctypes.data_as(C.c_void_p))
    assert ((u >= 0) & (u < 1)).all()


def test_strided_view_of_a_shared_stream(bind):
    """Row b of n rows that draw in turn from one generator (mt_rng.rs:22; draw order generation.rs:1055-1060) owns the
    elements b, b + n, ... of its stream."""
    u = np.zeros(60, np.float32)
    bind.lib().orc_mt_uniform(42, 60, u.ctypes.data_as(C.c_void_p))
    for n in (1, 3, 4):
        for b in range(n):
            v = np.zeros(60 // n, np.float32)
            bind.lib().orc_mt_uniform_strided(42, 60 // n, n, b, v.ctypes.data_as(C.c_void_p))
            assert np.array_equal(v, u[b::n][:60 // n])


def test_temperature(bind, kats):
    k = kats["temperature"]
    for c in k["cases"]:
        assert _proc(bind.lib(), 1, k["logits"], c["t"]).tolist() == c["expected"]


def test_top_k(bind, kats):
    k = kats["top_k"]
    assert np.array_equal(_proc(bind.lib(), 2, k["logits"], k["k"]), _vals(k["expected"]))


def test_repetition_penalty(bind, kats):
    k = kats["repetition_penalty"]
    assert _proc(bind.lib(), 0, k["logits"], k["penalty"], k["history"]).tolist() == k["expected"]
    # a token seen twice is penalised twice (sequential /p or *p per occurrence)
    out = _proc(bind.lib(), 0, [2.0, -1.0, 3.0], 2.0, [0, 0, 1])
    assert out.tolist() == [0.5, -2.0, 3.0]


def test_greedy(bind, kats):
    k = kats["greedy"]
    assert _sample(bind, k["logits"]) == k["expected"]


def test_suppress(bind, kats):
    k = kats["suppress_tokens"]
    assert np.array_equal(_proc(bind.lib(), 4, k["logits"], 0.0, k["suppress"]), _vals(k["expected"]))
    # the generate loop suppresses a contiguous id range except EOS (nn/generation.rs:848-853)
    assert _sample(bind, k["greedy_logits"], suppress_lo=1, suppress_hi=2) == k["greedy_expected"]


def test_min_new_tokens(bind, kats):
    k = kats["min_new_tokens"]
    for c in k["cases"]:
        s = bind.sampling_array(dict(do_sample=False, temperature=1.0, top_k=0, repetition_penalty=1.0,
                                     min_new_tokens=k["min_new_tokens"], eos_token_id=k["eos"]))
        l = np.array(k["logits"], np.float32)
        h = np.array(c["history"], np.int64)
        got = bind.lib().orc_sample_token(l.ctypes.data_as(C.c_void_p), len(l), s.ctypes.data_as(C.c_void_p),
                                          h.ctypes.data_as(C.c_void_p), len(h), 0)
        assert got == c["expected"]


def test_rotate_half(bind, kats):
    k = kats["rotate_half"]
    x = np.array(k["x"], np.float32)
    out = np.zeros_like(x)
    bind.lib().orc_rotate_half(x.ctypes.data_as(C.c_void_p), len(x), out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == k["expected"]


def test_hann(bind, kats):
    k = kats["hann_window"]
    w = np.zeros(k["n"], np.float32)
    bind.lib().orc_hann(k["n"], w.ctypes.data_as(C.c_void_p))
    for i, v in k["checks"]:
        assert abs(w[i] - v) < k["tol"]


def test_reflect_pad(bind, kats):
    k = kats["reflect_pad"]
    x = np.array(k["signal"], np.float32)
    out = np.zeros(len(x) + k["left"] + k["right"], np.float32)
    n = bind.lib().orc_reflect_pad(x.ctypes.data_as(C.c_void_p), len(x), k["left"], k["right"], out.ctypes.data_as(C.c_void_p))
    assert n == len(out) and out.tolist() == k["expected"]


def test_mel_filterbank(bind, kats):
    k = kats["mel_filterbank"]
    fb = np.zeros(k["shape"], np.float32)
    bind.lib().orc_mel_filterbank(k["n_fft"], k["num_mels"], k["sample_rate"], C.c_double(k["fmin"]), C.c_double(k["fmax"]),
                                  fb.ctypes.data_as(C.c_void_p))
    for r, c in k["positive"]:
        assert fb[r, c] > 0
    assert (fb >= 0).all()
    assert np.abs(fb[0, :3] - np.array(k["librosa_row0_first3"])).max() < k["librosa_tol"]
    assert np.abs(fb[1, :4] - np.array(k["librosa_row1_first4"])).max() < k["librosa_tol"]


def test_mel_shape(bind):
    audio = np.sin(np.arange(24000, dtype=np.float32) * 0.01)
    from oracle.bind import Oracle
    mel = Oracle().mel(audio)
    assert mel.shape == ((24000 + 768 - 1024) // 256 + 1, 128)
    assert np.isfinite(mel).all() and mel.min() >= np.log(1e-5) - 1e-6


def test_position_ids_with_padding(bind, kats):
    k = kats["position_ids_with_padding"]
    lens = np.array(k["lengths"], np.int32)
    out = np.zeros((len(lens), k["max_len"]), np.int64)
    bind.lib().orc_position_ids_with_padding(lens.ctypes.data_as(C.c_void_p), len(lens), k["max_len"], out.ctypes.data_as(C.c_void_p))
    assert out.tolist() == k["expected"]


def test_attention_mask_from_lengths(bind, kats):
    k = kats["attention_mask_from_lengths"]
    lens = np.array(k["lengths"], np.int32)
    m = np.zeros((len(lens), k["max_len"], k["max_len"]), np.float32)
    bind.lib().orc_attention_mask_from_lengths(lens.ctypes.data_as(C.c_void_p), len(lens), k["max_len"], m.ctypes.data_as(C.c_void_p))
    for b, i, j, v in k["checks"]:
        assert m[b, i, j] == (-np.inf if v == "-inf" else v)
