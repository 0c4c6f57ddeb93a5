"""Continuous batching (qtts_generate_continuous): utterances go through a few row slots, a finished row's slot is handed to the
next queued utterance while the others keep decoding.  Every utterance must get exactly what a qtts_generate call with that
request alone returns (frame count, effective length, code ids).  The synthetic codec head has its EOS row scaled up so that
sampled runs end at scattered frames."""
import numpy as np
import pytest

from conftest import SEED

pytestmark = pytest.mark.gpu


def _model(dtype, eos_gain=6.0):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set("T0", "custom_voice")
    eos = cfg["talker_config"]["codec_eos_token_id"]
    m = Model(cfg, None, LoaderConfig(dtype=dtype))
    for k, t in synth.iter_weights_torch(synth.talker_keys(cfg), SEED, "cuda:0", dtype == "bf16"):
        if k == "talker.codec_head.weight":
            t = t.clone()
            t[eos] *= eos_gain          # a wider EOS logit: drawn now and then, at a frame that depends on the row's history
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    return m, tmpl


def _requests(tmpl, n):
    from qwen3_tts_rs_amd import GenerationRequest, synth
    return [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 5 + (7 * u) % 23, u), language="english", speaker="speaker1",
                              seed=900 + u) for u in range(n)]


@pytest.mark.parametrize("use_graph", [True, False])
def test_every_utterance_matches_its_batch_of_one(use_graph):
    from qwen3_tts_rs_amd import SamplingConfig
    m, tmpl = _model("f32")
    try:
        n, max_new = 14, 40
        reqs = _requests(tmpl, n)
        sampling = SamplingConfig(temperature=1.0, top_k=0)
        alone = [m.generate([r], sampling, max_new) for r in reqs]
        lens = [o.num_steps for o in alone]
        assert len(set(lens)) >= 4 and min(lens) < max_new, f"the fixture should end rows at scattered frames, got {lens}"
        for slots in (4, 1, 32):
            out = m.generate(reqs, sampling, max_new, slots=slots, use_graph=use_graph)
            for u in range(n):
                a = alone[u]
                assert out.steps_per_utterance[u] == a.num_steps, f"slots {slots} utterance {u}"
                assert out.effective_lengths[u] == a.effective_lengths[0], f"slots {slots} utterance {u}"
                assert np.array_equal(out.codes[u, :a.num_steps], a.codes[0]), f"slots {slots} utterance {u}"
    finally:
        m.close()


def test_greedy_rows_and_plain_batch_agree_up_to_each_effective_length():
    """bf16, greedy, EOS as the model produces it: the continuous run's codes equal the plain batch's rows up to every row's own
    end (the plain batch keeps all rows running until the last one is done)."""
    from qwen3_tts_rs_amd import SamplingConfig
    m, tmpl = _model("bf16", eos_gain=12.0)
    try:
        n, max_new = 9, 30
        reqs = _requests(tmpl, n)
        greedy = SamplingConfig.cli_greedy()
        plain = m.generate(reqs, greedy, max_new)
        cont = m.generate(reqs, greedy, max_new, slots=3)
        for u in range(n):
            e = plain.effective_lengths[u]
            assert cont.effective_lengths[u] == e
            assert np.array_equal(cont.codes[u, :e], plain.codes[u, :e])
        with pytest.raises(Exception):
            m.generate(reqs, greedy, max_new, slots=3, trace=True)
    finally:
        m.close()
