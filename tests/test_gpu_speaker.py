"""GPU parity of the mel front end and the ECAPA-TDNN speaker encoder through the C ABI."""
import numpy as np
import pytest

from conftest import make_oracle
from golden.make_golden import test_audio as make_audio

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def base_f32():
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", model_type="base", codec=False)
    yield m, cfg
    m.close()


def test_mel_matches_golden_and_oracle(base_f32, bind, golden):
    m, cfg = base_f32
    clips = [make_audio(24000, 1), make_audio(5000, 2), make_audio(36111, 3), make_audio(1024, 4)]
    mels = m.compute_mel_spectrogram(clips)
    assert mels[0].shape == golden["mel_1s"].shape
    # f32 FFT on the device (the reference's rustfft is f32 as well) against the oracle's f64 FFT: the log amplifies
    # rounding in near-silent bins, so the bound is relative in the linear domain and loose in the log domain
    def close(g, ref):
        return np.abs(g - ref).max() < 1e-2 and (np.abs(np.exp(g) - np.exp(ref)) <= 2e-6 + 2e-4 * np.exp(ref)).all()
    assert close(mels[0], golden["mel_1s"])
    o = bind.Oracle()
    for a, g in zip(clips, mels):
        ref = o.mel(a)
        assert g.shape == ref.shape == ((len(a) + 768 - 1024) // 256 + 1, 128)
        assert close(g, ref)


def test_mel_rejects_short_audio(base_f32):
    from qwen3_tts_rs_amd import QttsError
    m, cfg = base_f32
    with pytest.raises(QttsError):
        m.compute_mel_spectrogram([np.zeros(300, np.float32)])


def test_speaker_embedding_matches_oracle(base_f32, bind, golden):
    m, cfg = base_f32
    ob, _, _, _ = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=False)
    clips = [make_audio(24000, 1), make_audio(9000, 2), make_audio(30000, 3)]
    emb = m.extract_speaker_embedding(clips)
    assert emb.shape == (3, cfg["speaker_encoder_config"]["enc_dim"])
    assert np.abs(emb[0] - golden["spk_embed_1s"]).max() < 2e-3 * max(1.0, np.abs(golden["spk_embed_1s"]).max())
    for a, e in zip(clips, emb):
        ref = ob.speaker_embed(ob.mel(a))
        assert np.abs(e - ref).max() < 2e-3 * max(1.0, np.abs(ref).max())
    # a clip's embedding does not depend on what else is in the batch
    alone = m.extract_speaker_embedding([clips[1]])
    assert np.abs(alone[0] - emb[1]).max() < 1e-5


def test_speaker_embedding_from_a_mel_equals_the_audio_path(base_f32):
    """create_voice_clone_prompt_from_mel (model/mod.rs:579-612): the caller's mel through the same encoder."""
    m, cfg = base_f32
    clips = [make_audio(24000, 3), make_audio(17000, 4)]
    mels = m.compute_mel_spectrogram(clips)
    a = m.extract_speaker_embedding(clips)
    b = m.encode_speaker_from_mel(mels)
    assert np.abs(a - b).max() < 1e-5
    item = m.create_voice_clone_prompt_from_mel(mels[1])
    assert item.x_vector_only_mode and np.abs(item.ref_spk_embedding - a[1]).max() < 1e-5


def test_speaker_embedding_matches_an_independent_ecapa_implementation(base_f32):
    """The device encoder against transformers' ECAPA_TimeDelayNet on the same weights and mel inputs
    (tests/golden/make_ecapa_golden.py, committed as tests/golden/ecapa_golden.npz)."""
    import os
    m, cfg = base_f32
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ecapa_golden.npz"))
    e = m.encode_speaker_from_mel([g["mel_a"], g["mel_b"]])
    for i, name in enumerate(("a", "b")):
        assert np.abs(e[i] - g["emb_" + name]).max() < 1e-4, name


def test_speaker_embedding_feeds_generation(base_f32, bind):
    """VoiceClone (x-vector only): embedding -> one prompt position -> tokens identical to the oracle fed the oracle's embedding."""
    from golden.make_golden import GREEDY
    from qwen3_tts_rs_amd import SamplingConfig, synth
    m, cfg = base_f32
    ob, _, _, tmpl = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=False)
    clip = make_audio(20000, 5)
    emb = m.extract_speaker_embedding([clip])[0]
    text = synth.synthetic_text_ids(tmpl, 11, 4)
    res = m.generate_voice_clone_full(text, "english", emb, options=dict(do_sample=False, subtalker_do_sample=False, temperature=1.0,
                                                                           max_new_tokens=10), decode=False)
    r = ob.generate(text, bind.resolve_codec_prefill(cfg, "english", None), GREEDY, 10, speaker_embed=emb)
    assert np.array_equal(res.codes[0], r["codes"])


def test_speaker_encoder_missing_on_custom_voice_model():
    from qwen3_tts_rs_amd import QttsError, synthetic_model
    m, cfg, _, _ = synthetic_model("T0", dtype="f32", model_type="custom_voice", codec=False)
    try:
        with pytest.raises(QttsError, match="Speaker encoder not available"):
            m.extract_speaker_embedding([make_audio(24000, 1)])
    finally:
        m.close()


def test_bf16_speaker_embedding(bind):
    from qwen3_tts_rs_amd import synthetic_model
    m, cfg, _, _ = synthetic_model("T0", dtype="bf16", model_type="base", codec=False, round_to_bf16=True)
    ob, _, _, _ = make_oracle(bind, "T0", "bf16", model_type="base", talker=True, codec=False)
    try:
        clip = make_audio(24000, 1)
        e = m.extract_speaker_embedding([clip])[0]
        ref = ob.speaker_embed(ob.mel(clip))
        cos = float(np.dot(e, ref) / (np.linalg.norm(e) * np.linalg.norm(ref)))
        assert cos > 0.995
    finally:
        m.close()


def test_full_size_speaker_encoder_properties():
    """Reference-default ECAPA shapes (512-channel blocks, 1536-wide pooling), bf16, a ragged batch of 3-10 s clips:
    finite output, batch-composition independence, determinism."""
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    from conftest import SEED
    cfg, ccfg, tmpl = synth.shape_set("S06", "base")
    m = Model(cfg, None, LoaderConfig(dtype="bf16"))
    for k, t in synth.iter_weights_torch(synth.speaker_keys(cfg), SEED, "cuda:0", True):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    try:
        clips = [make_audio(72000 + 17001 * i, i) for i in range(10)]
        e = m.extract_speaker_embedding(clips)
        assert e.shape == (10, cfg["speaker_encoder_config"]["enc_dim"]) and np.isfinite(e).all()
        e2 = m.extract_speaker_embedding(clips[::-1])[::-1]
        assert np.abs(e - e2).max() <= 2e-2 * np.abs(e).max()
        assert np.array_equal(e, m.extract_speaker_embedding(clips))
        assert np.abs(e[0] - e[9]).max() > 1e-3
    finally:
        m.close()


def test_icl_voice_clone_end_to_end(bind, golden):
    """ICL voice clone: reference codes + transcript in the prompt, decode of [ref_code || generated], proportional cut of the
    reference part (model/generate.rs:545-634)."""
    from conftest import snr_db
    from golden.make_golden import GREEDY
    from qwen3_tts_rs_amd import VoiceClonePromptItem, synthetic_model
    m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", model_type="base", codec=True)
    ob, _, _, _ = make_oracle(bind, "T0", "f32", model_type="base", talker=True, codec=True)
    try:
        emb = m.extract_speaker_embedding([make_audio(20000, 6)])[0]
        prompt = VoiceClonePromptItem.icl(golden["ref_codes"], emb, golden["ref_text_ids"])
        res = m.generate_voice_clone_full(golden["text_ids"], "english", prompt,
                                          options=dict(do_sample=False, subtalker_do_sample=False, temperature=1.0, max_new_tokens=12))
        r = ob.generate(golden["text_ids"], bind.resolve_codec_prefill(cfg, "english", None), GREEDY, 12, speaker_embed=emb,
                        ref_text_ids=golden["ref_text_ids"], ref_codes=golden["ref_codes"], non_streaming=True)
        assert np.array_equal(res.codes[0], r["codes"])
        comb = np.concatenate([golden["ref_codes"], r["codes"][:r["effective_len"]]])
        full = ob.codec_decode(comb)
        cut = int(len(golden["ref_codes"]) / len(comb) * len(full))
        assert len(res.audio[0]) == len(full) - cut
        assert snr_db(full[cut:], res.audio[0]) >= 40.0
        with pytest.raises(Exception, match="ICL mode requires ref_text"):
            m.generate_voice_clone_full(golden["text_ids"], "english", VoiceClonePromptItem(emb, golden["ref_codes"], False, None))
    finally:
        m.close()
