"""File formats shared with the reference (voice prompts, code files, WAV) and the safetensors model-directory loader."""
import base64
import json
import os
import struct
import wave

import numpy as np
import pytest


def test_voice_prompt_round_trip_and_layout(tmp_path):
    from qwen3_tts_rs_amd import VoiceClonePromptItem
    from qwen3_tts_rs_amd.io import load_voice_prompt, save_voice_prompt
    from safetensors.numpy import load
    emb = np.linspace(-1, 1, 128, dtype=np.float32)
    code = np.arange(36, dtype=np.uint32).reshape(9, 4)
    p = str(tmp_path / "voice.json")
    save_voice_prompt(VoiceClonePromptItem.icl(code, emb, [1, 2, 3]), p, x_vector_only=False, ref_text="hello there")
    doc = json.load(open(p))
    assert set(doc) == {"version", "x_vector_only", "ref_text", "prompt_data"} and doc["version"] == 1
    t = load(base64.b64decode(doc["prompt_data"]))
    assert t["ref_spk_embedding"].dtype == np.float32 and t["ref_code"].dtype == np.int64 and t["ref_code"].shape == (9, 4)
    item = load_voice_prompt(p, ref_text_ids=[1, 2, 3])
    assert item.is_icl() and np.array_equal(item.ref_code, code) and np.array_equal(item.ref_spk_embedding, emb) and item.ref_text == "hello there"
    # x-vector only: no codes, no transcript
    save_voice_prompt(VoiceClonePromptItem.x_vector_only(emb), p, x_vector_only=True)
    item = load_voice_prompt(p)
    assert not item.is_icl() and item.ref_code is None
    doc = json.load(open(p))
    doc["version"] = 2
    json.dump(doc, open(p, "w"))
    with pytest.raises(Exception, match="version mismatch"):
        load_voice_prompt(p)


def test_codes_file_round_trip(tmp_path):
    from qwen3_tts_rs_amd.io import load_codes, save_codes
    codes = np.random.default_rng(0).integers(0, 2048, (11, 16)).astype(np.uint32)
    p = str(tmp_path / "codes.json")
    save_codes(codes, p, 24000)
    doc = json.load(open(p))
    assert (doc["version"], doc["sample_rate"], doc["num_codebooks"], doc["num_steps"]) == (1, 24000, 16, 11)
    back, sr = load_codes(p)
    assert sr == 24000 and np.array_equal(back, codes)


def test_wav_writer_truncates_like_the_reference(tmp_path):
    from qwen3_tts_rs_amd.io import write_wav
    x = np.array([0.0, 0.5, -0.5, 1.5, -1.5, 0.99999, -3.05e-5, 3.05e-5], np.float32)
    p = str(tmp_path / "a.wav")
    write_wav(p, x, 24000)
    with wave.open(p) as w:
        assert (w.getnchannels(), w.getsampwidth(), w.getframerate(), w.getnframes()) == (1, 2, 24000, len(x))
        pcm = np.frombuffer(w.readframes(len(x)), "<i2")
    # `(sample * 32767.0) as i16` truncates toward zero after the clamp
    assert pcm.tolist() == [0, 16383, -16383, 32767, -32767, 32766, 0, 0]
    with pytest.raises(Exception, match="0 samples"):
        write_wav(p, np.zeros(0, np.float32), 24000)


@pytest.mark.gpu
def test_model_loader_reads_a_safetensors_directory(tmp_path, bind, golden):
    """ModelLoader::from_local_dir + load_tts_model on a directory laid out like a released checkpoint
    (config.json, generate_config.json, model.safetensors, speech_tokenizer/{config.json, model.safetensors})."""
    from conftest import SEED, make_oracle
    from golden.make_golden import GREEDY
    from qwen3_tts_rs_amd import LoaderConfig, ModelLoader, synth
    from safetensors.numpy import save_file
    cfg, ccfg, tmpl = synth.shape_set("T0")
    d = tmp_path / "model"
    (d / "speech_tokenizer").mkdir(parents=True)
    json.dump(cfg, open(d / "config.json", "w"))
    json.dump(ccfg, open(d / "speech_tokenizer" / "config.json", "w"))
    json.dump(dict(do_sample=False, subtalker_dosample=False, temperature=1.0, max_new_tokens=16), open(d / "generate_config.json", "w"))
    save_file({k: a for k, a in synth.iter_weights(synth.talker_keys(cfg), SEED)}, str(d / "model.safetensors"))
    ck = {k: a for k, a in synth.iter_weights(synth.codec_keys(ccfg), SEED)}
    ck["encoder.ignored.weight"] = np.zeros((4, 4), np.float32)      # codec encoder tensors are skipped
    save_file(ck, str(d / "speech_tokenizer" / "model.safetensors"))
    loader = ModelLoader.from_local_dir(str(d))
    assert loader.generate_defaults.subtalker_do_sample is False and loader.generate_defaults.max_new_tokens == 16
    m = loader.load_tts_model(LoaderConfig(dtype="f32"))
    try:
        res = m.generate_custom_voice_full(golden["text_ids"], "speaker1", "english")
        assert res.sample_rate == 24000 and res.codes.shape[1] == 16
        assert np.array_equal(res.codes[0], golden["greedy_codes"][:16])       # generate_config.json drove the sampling
        assert len(res.audio[0]) == 96 * res.effective_lengths[0] - 38 or len(res.audio[0]) > 0
    finally:
        m.close()
    with pytest.raises(FileNotFoundError):
        ModelLoader.from_local_dir(str(tmp_path / "nowhere"))
