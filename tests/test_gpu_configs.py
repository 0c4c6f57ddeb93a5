"""The benchmark configurations at their real shapes and batch sizes (the headline run at full length, the others with
shortened generation lengths) through size-independent properties of the algorithm: determinism, batch-composition
independence (per-row logit bounds), length arithmetic, value ranges.  The comparisons with the CPU oracle at these shapes
are in test_gpu_parity_full_size.py."""
import numpy as np
import pytest

from conftest import SEED
from golden.make_golden import GREEDY, test_audio as make_audio

pytestmark = pytest.mark.gpu

BF16_LOGIT_TOL = 2e-2


def _logits_agree_while_histories_do(codes_a, logits_a, codes_b, logits_b):
    """max-abs logit difference of one utterance between two runs, over the frames whose code histories are equal."""
    same = (codes_a == codes_b).all(axis=1)
    n = len(same)
    upto = min((int(np.argmin(same)) if not same.all() else n) + 1, n)
    return float(np.abs(logits_a[:upto] - logits_b[:upto]).max())


def _model(shape, model_type, talker=True, codec=False, speaker=False, dtype="bf16"):
    import torch
    from qwen3_tts_rs_amd import LoaderConfig, Model, synth
    cfg, ccfg, tmpl = synth.shape_set(shape, model_type)
    m = Model(cfg, ccfg if codec else None, LoaderConfig(dtype=dtype))
    keys = (synth.talker_keys(cfg) if talker else []) + (synth.speaker_keys(cfg) if speaker else []) + (synth.codec_keys(ccfg) if codec else [])
    for k, t in synth.iter_weights_torch(keys, SEED, "cuda:0", dtype == "bf16"):
        m.load_tensor(k, t)
    m.finalize()
    torch.cuda.empty_cache()
    return m, cfg, ccfg, tmpl


def _greedy():
    from qwen3_tts_rs_amd import SamplingConfig
    return SamplingConfig(**{k: GREEDY[k] for k in SamplingConfig().__dict__ if k in GREEDY})


def test_small_model_batch_one_decode_and_codec():
    """0.6B-like Base model, batch 1: decode + codec decode end to end."""
    from qwen3_tts_rs_amd import GenerationRequest, synth
    m, cfg, ccfg, tmpl = _model("S06", "base", codec=True)
    try:
        text = synth.synthetic_text_ids(tmpl, 12, 1)
        xvec = synth.make_tensor("cfg2.xvector", (cfg["talker_config"]["hidden_size"],), "normal:1.0", SEED)
        req = GenerationRequest(text_ids=text, language="english", speaker_embed=xvec)
        a = m.generate([req], _greedy(), 40)
        b = m.generate([req], _greedy(), 40, use_graph=False)
        # (a random-init model may reach EOS before the 40 frames: the length is whatever the greedy path decides)
        assert np.array_equal(a.codes, b.codes) and a.num_steps == b.num_steps and 2 <= a.num_steps <= 40
        assert a.effective_lengths == b.effective_lengths and a.effective_lengths[0] >= 2
        wav = m.decode_output(a)[0]
        assert len(wav) == 1920 * a.effective_lengths[0] - 555 and np.isfinite(wav).all() and np.abs(wav).max() <= 1.0
        # the same utterance inside a batch of other utterances: its logits stay within the bf16 tolerance for as long as
        # the code histories agree (the batched GEMM path sums in another order)
        others = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 9 + i, 10 + i), language="english", speaker_embed=xvec) for i in range(3)]
        a4 = m.generate([req], _greedy(), 4, trace=True)
        c = m.generate([req] + others, _greedy(), 4, trace=True)
        assert _logits_agree_while_histories_do(a4.codes[0], a4.logits[0], c.codes[0], c.logits[0]) <= BF16_LOGIT_TOL
    finally:
        m.close()


def test_voice_clone_batch_256():
    """1.7B-like Base model, 256 reference clips: mel + ECAPA on the device, x-vector prompts, batched decode."""
    from qwen3_tts_rs_amd import GenerationRequest, synth
    m, cfg, ccfg, tmpl = _model("S17", "base", speaker=True)
    try:
        n = 256
        clips = [make_audio(72000 + 641 * (i % 257), i % 11) for i in range(n)]     # 3.0 - 9.9 s at 24 kHz
        emb = m.extract_speaker_embedding(clips)
        assert emb.shape == (n, cfg["talker_config"]["hidden_size"]) and np.isfinite(emb).all()
        again = m.extract_speaker_embedding(clips[:7])
        assert np.abs(again - emb[:7]).max() <= 2e-2 * np.abs(emb[:7]).max()
        reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 8 + i % 40, i), language="english", speaker_embed=emb[i])
                for i in range(n)]
        out = m.generate(reqs, _greedy(), 3, trace=True)
        assert out.codes.shape == (n, 3, cfg["talker_config"]["num_code_groups"])
        t = cfg["talker_config"]
        assert (out.codes[:, :, 0] < t["vocab_size"]).all() and (out.codes[:, :, 1:] < t["code_predictor_config"]["vocab_size"]).all()
        sub = m.generate(reqs[64:128], _greedy(), 3, trace=True)     # rows do not depend on their batch
        for i in range(64):
            d = _logits_agree_while_histories_do(sub.codes[i], sub.logits[i], out.codes[64 + i], out.logits[64 + i])
            assert d <= BF16_LOGIT_TOL, f"row {64 + i}: {d:.4f}"
    finally:
        m.close()


def test_long_form_codec_decode():
    """Codec standalone, 10-minute sequences (7500 frames): decode_output's policy chunks them (256 / 32)."""
    m, cfg, ccfg, tmpl = _model("S17", "custom_voice", talker=False, codec=True)
    try:
        rng = np.random.default_rng(3)
        codes = rng.integers(0, 2048, (3, 7500, 16)).astype(np.uint32)
        wavs = m.decode_codes(codes, [7500, 6000, 513], chunk=-2)
        for w, T in zip(wavs, (7500, 6000, 513)):
            nchunks = -(-T // 256)
            assert len(w) == 1920 * T - 555 * nchunks
            assert np.isfinite(w).all() and np.abs(w).max() <= 1.0
        # chunk i depends only on its own frames and 32 frames of left context
        part = m.decode_codes(codes[:1, 224:512], chunk=0)[0]          # second window of row 0 with its context
        ref = wavs[0][1920 * 256 - 555:][:1920 * 256 - 555]
        n = min(len(ref), len(part) - 32 * 1920)
        from conftest import snr_db
        assert snr_db(part[32 * 1920:32 * 1920 + n], ref[:n]) > 36.0
    finally:
        m.close()


def test_bench_prints_one_contract_line():
    """bench.py end to end at a reduced size: one JSON line with the fields the harness reads, the roofline of the decode frame
    measured in the run and the CPU port timed beside it."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--frames", "30",
                        "--pipeline", "2", "--cpu-frames", "4"], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().split("\n") if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"] and d["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]


def test_headline_configuration_at_full_size():
    """The benchmark's own workload (1.7B-like shapes, bf16, 64 utterances, 375 frames, seeded sampling, codec decode) through
    size-independent properties: two runs are identical code for code, the hipGraph replay equals kernel-by-kernel launches, every
    code is inside its codebook, the waveform has 1920 T - 555 samples in [-1, 1], and the first 25 frames pushed through a
    streaming session are the head of the whole decode."""
    from conftest import snr_db
    from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synth
    m, cfg, ccfg, tmpl = _model("S17", "custom_voice", codec=True)
    try:
        B, T = 64, 375
        reqs = [GenerationRequest(text_ids=synth.synthetic_text_ids(tmpl, 8 + (u * 37 + 11) % 57, u), language="english", speaker="speaker1",
                                  seed=42 + u) for u in range(B)]
        a = m.generate(reqs, SamplingConfig(), T, min_new_tokens=T)
        b = m.generate(reqs, SamplingConfig(), T, min_new_tokens=T)
        assert a.num_steps == T and a.effective_lengths == [T] * B
        assert np.array_equal(a.codes, b.codes)
        c = m.generate(reqs[:8], SamplingConfig(), 40, min_new_tokens=40, use_graph=False)
        d = m.generate(reqs[:8], SamplingConfig(), 40, min_new_tokens=40)
        assert np.array_equal(c.codes, d.codes)
        t = cfg["talker_config"]
        assert (a.codes[:, :, 0] < t["vocab_size"] - 1024).all() and (a.codes[:, :, 1:] < t["code_predictor_config"]["vocab_size"]).all()
        wavs = m.decode_output(a)
        for w in wavs:
            assert len(w) == 1920 * T - 555 and np.isfinite(w).all() and np.abs(w).max() <= 1.0
        Q = ccfg["decoder_config"]["num_quantizers"]
        with m.open_codec_stream(B, 25) as s:
            head = s.push(a.codes[:, :25, :Q])
        assert head.shape == (B, 1920 * 25 - 555)
        for u in (0, 31, 63):
            assert snr_db(wavs[u][:head.shape[1]], head[u]) > 36.0, u
    finally:
        m.close()
