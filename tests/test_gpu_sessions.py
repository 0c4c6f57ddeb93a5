"""Decode sessions (qtts_session_open): several batches in flight on one set of weights.  A session's results equal the
model's own on the same requests, whether it runs alone or while other sessions run on other host threads."""
import threading

import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


def _model(dtype="bf16"):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set("T0", "custom_voice")
    m = Model(cfg, ccfg, LoaderConfig(dtype=dtype))
    keys = synth.talker_keys(cfg) + synth.codec_keys(ccfg)
    for k, t in synth.iter_weights_torch(keys, SEED, "cuda:0", dtype == "bf16"):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    return m, tmpl


def _requests(tmpl, n, salt):
    from qwen3_tts_rs_amd import GenerationRequest, synth
    return [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 6 + (3 * u + salt) % 9, u + 17 * salt), language="english",
                              speaker="speaker1", seed=1000 + u + 100 * salt) for u in range(n)]


def test_sessions_share_weights_and_reproduce_the_model():
    from qwen3_tts_rs_amd import SamplingConfig
    m, tmpl = _model()
    try:
        sessions = [m.open_session() for _ in range(2)]
        sampling = SamplingConfig()
        frames = 24
        batches = [_requests(tmpl, 5, s) for s in range(3)]
        ref = [m.generate(b, sampling, frames) for b in batches]
        ref_wavs = [m.decode_output(o) for o in ref]
        # one after the other: a session alone
        for ses, b, r in zip(sessions, batches[1:], ref[1:]):
            out = ses.generate(b, sampling, frames)
            assert np.array_equal(out.codes, r.codes) and list(out.effective_lengths) == list(r.effective_lengths)
        # all three at once, each on its own host thread, generate + codec decode, twice in a row
        runners = [m] + sessions
        got = [None] * 3
        errs = []

        def work(i):
            try:
                for _ in range(2):
                    out = runners[i].generate(batches[i], sampling, frames)
                    got[i] = (out, runners[i].decode_output(out))
            except Exception as ex:  # surfaced after the join
                errs.append(ex)

        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for i in range(3):
            assert np.array_equal(got[i][0].codes, ref[i].codes), f"stream {i}"
            for a, b in zip(got[i][1], ref_wavs[i]):
                assert np.array_equal(a, b), f"stream {i}: waveform"
        for ses in sessions:
            ses.close()
    finally:
        m.close()


def test_session_lifecycle_errors():
    from qwen3_tts_rs_amd import LoaderConfig, Model, QttsError, synth
    cfg, ccfg, tmpl = synth.shape_set("T0", "custom_voice")
    m = Model(cfg, None, LoaderConfig(dtype="f32"))
    try:
        with pytest.raises(QttsError):
            m.open_session()          # not finalized yet
    finally:
        m.close()
    m, tmpl = _model("f32")
    try:
        ses = m.open_session()
        with pytest.raises(QttsError):
            ses.load_tensor("talker.model.norm.weight", np.zeros(8, np.float32))   # a session cannot take weights
        ses.close()
        ses.close()                   # idempotent
    finally:
        m.close()
