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


def _write_model_dir(d, cfg, ccfg, tensors_main, tensors_codec, main_name="model.safetensors"):
    from safetensors.torch import save_file
    (d / "speech_tokenizer").mkdir(parents=True)
    json.dump(cfg, open(d / "config.json", "w"))
    json.dump(ccfg, open(d / "speech_tokenizer" / "config.json", "w"))
    json.dump(dict(do_sample=False, subtalker_dosample=False, temperature=1.0, max_new_tokens=16), open(d / "generate_config.json", "w"))
    save_file(tensors_main, str(d / main_name))
    save_file(tensors_codec, str(d / "speech_tokenizer" / "model.safetensors"))


@pytest.mark.gpu
def test_native_directory_loader_matches_the_python_one(tmp_path, golden):
    """qtts_model_load_dir (the C++ file reader a Rust caller would use) on checkpoint-shaped directories: F32 tensors give the
    golden greedy tokens and the Python loader's waveform; a first-shard file name is found like model.safetensors; BF16- and
    F16-stored tensors load into a bf16 model exactly as their f32 values would; missing or cut files are I/O errors."""
    import torch
    from conftest import SEED
    from qwen3_tts_rs_amd import LoaderConfig, ModelLoader, QttsError, synth
    cfg, ccfg, tmpl = synth.shape_set("T0")
    main = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in synth.iter_weights(synth.talker_keys(cfg), SEED)}
    codec = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in synth.iter_weights(synth.codec_keys(ccfg), SEED)}
    codec["encoder.ignored.weight"] = torch.zeros(4, 4)
    main["talker.model.bookkeeping.num_batches_tracked"] = torch.zeros(1, dtype=torch.int64)   # non-float tensors are skipped
    _write_model_dir(tmp_path / "f32", cfg, ccfg, main, codec, main_name="model-00001-of-00002.safetensors")
    loader = ModelLoader.from_local_dir(str(tmp_path / "f32"))
    a = loader.load_tts_model(LoaderConfig(dtype="f32"))
    b = loader.load_tts_model_native(LoaderConfig(dtype="f32"))
    try:
        ra = a.generate_custom_voice_full(golden["text_ids"], "speaker1", "english")
        rb = b.generate_custom_voice_full(golden["text_ids"], "speaker1", "english")
        assert np.array_equal(ra.codes, rb.codes) and list(ra.effective_lengths) == list(rb.effective_lengths)
        assert np.array_equal(ra.codes[0], golden["greedy_codes"][:16])
        assert np.array_equal(ra.audio[0], rb.audio[0])
    finally:
        a.close()
        b.close()
    # the same weights stored as BF16 / F16 (values that both formats hold exactly) against F32 storage, loaded into bf16 models
    main.pop("talker.model.bookkeeping.num_batches_tracked")
    exact = lambda t: t.to(torch.bfloat16).to(torch.float16).to(torch.float32) if t.dtype == torch.float32 else t
    both = lambda t: exact(exact(t))
    m16 = {k: both(t) for k, t in main.items()}
    c16 = {k: both(t) for k, t in codec.items()}
    ok = all(torch.equal(t.to(torch.bfloat16).to(torch.float32), t) and torch.equal(t.to(torch.float16).to(torch.float32), t) for t in m16.values())
    assert ok, "test setup: values must be exact in both 16-bit formats"
    _write_model_dir(tmp_path / "s32", cfg, ccfg, m16, c16)
    _write_model_dir(tmp_path / "sbf", cfg, ccfg, {k: t.to(torch.bfloat16) for k, t in m16.items()}, {k: t.to(torch.bfloat16) for k, t in c16.items()})
    _write_model_dir(tmp_path / "sf16", cfg, ccfg, {k: t.to(torch.float16) for k, t in m16.items()}, {k: t.to(torch.float16) for k, t in c16.items()})
    outs = []
    for name in ("s32", "sbf", "sf16"):
        m = ModelLoader.from_local_dir(str(tmp_path / name)).load_tts_model_native(LoaderConfig(dtype="bf16"))
        try:
            r = m.generate_custom_voice_full(golden["text_ids"], "speaker1", "english")
            outs.append((r.codes.copy(), r.audio[0].copy()))
        finally:
            m.close()
    for codes, audio in outs[1:]:
        assert np.array_equal(codes, outs[0][0]) and np.array_equal(audio, outs[0][1])
    # errors
    (tmp_path / "empty").mkdir()
    json.dump(cfg, open(tmp_path / "empty" / "config.json", "w"))
    with pytest.raises(QttsError) as ei:
        ModelLoader.from_local_dir(str(tmp_path / "empty")).load_tts_model_native(LoaderConfig(dtype="f32"))
    assert ei.value.code == -6 and "model.safetensors" in str(ei.value)
    cut = tmp_path / "cut"
    _write_model_dir(cut, cfg, ccfg, main, codec)
    raw = open(cut / "model.safetensors", "rb").read()
    open(cut / "model.safetensors", "wb").write(raw[: len(raw) // 2])
    with pytest.raises(QttsError) as ei:
        ModelLoader.from_local_dir(str(cut)).load_tts_model_native(LoaderConfig(dtype="f32"))
    assert ei.value.code == -6


@pytest.mark.gpu
def test_compiled_host_program_drives_the_c_abi(tmp_path, golden):
    """tools/qtts_synth (plain C++ on include/qtts.h: load_dir -> generate -> codec_decode -> WAV) gives the golden greedy tokens
    and, byte for byte, the WAV files the Python mirror writes for the same batch."""
    import subprocess
    import torch
    from conftest import ROOT, SEED
    from qwen3_tts_rs_amd import GenerationRequest, LoaderConfig, ModelLoader, SamplingConfig, synth
    from qwen3_tts_rs_amd.io import write_wav
    exe = os.path.join(ROOT, "tools", "qtts_synth")
    assert os.path.exists(exe), "tools/qtts_synth is built by make -C qwen3-tts-rs_amd/csrc"
    cfg, ccfg, tmpl = synth.shape_set("T0")
    main = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in synth.iter_weights(synth.talker_keys(cfg), SEED)}
    codec = {k: torch.from_numpy(np.ascontiguousarray(a)) for k, a in synth.iter_weights(synth.codec_keys(ccfg), SEED)}
    d = tmp_path / "model"
    _write_model_dir(d, cfg, ccfg, main, codec)
    texts = [np.asarray(golden["text_ids"]), synth.synthetic_text_ids(tmpl, 9, 1)]
    with open(tmp_path / "ids.txt", "w") as f:
        for t in texts:
            f.write(" ".join(str(int(v)) for v in t) + "\n")
    r = subprocess.run([exe, str(d), str(tmp_path / "ids.txt"), str(tmp_path / "out.wav"), "--speaker", "speaker1", "--language", "english",
                        "--dtype", "f32", "--max-new-tokens", "16", "--greedy", "--codes", str(tmp_path / "codes.txt")],
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    blocks = [b for b in open(tmp_path / "codes.txt").read().split("\n\n") if b.strip()]
    got = [np.array([[int(v) for v in line.split()] for line in b.strip().split("\n")], np.uint32) for b in blocks]
    m = ModelLoader.from_local_dir(str(d)).load_tts_model_native(LoaderConfig(dtype="f32"))
    try:
        out = m.generate([GenerationRequest(text_ids=t, language="english", speaker="speaker1") for t in texts], SamplingConfig.cli_greedy(), 16)
        wavs = m.decode_output(out)
    finally:
        m.close()
    assert np.array_equal(got[0], golden["greedy_codes"][:len(got[0])])
    for i in range(2):
        assert np.array_equal(got[i], out.codes[i][:out.effective_lengths[i]])
        write_wav(str(tmp_path / f"py.{i}.wav"), wavs[i], 24000)
        assert open(tmp_path / f"out.{i}.wav", "rb").read() == open(tmp_path / f"py.{i}.wav", "rb").read()
    # errors surface with the library's message and a non-zero status
    r = subprocess.run([exe, str(tmp_path / "nowhere"), str(tmp_path / "ids.txt"), str(tmp_path / "x.wav")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "config.json not found" in r.stderr
