"""Alternative kernel schedules selected by environment switches stay correct: every variant must reproduce the golden
tokens / waveforms.  The switches are read once per process, so each variant runs in a child process."""
import os
import subprocess
import sys

import numpy as np

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SCRIPT = r"""
import sys, os
import numpy as np
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import snr_db
from golden.make_golden import GREEDY, SAMPLED
from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synthetic_model
g = np.load(os.path.join(ROOT, "tests", "golden", "t0_golden.npz"))
m, cfg, ccfg, tmpl = synthetic_model("T0", dtype="f32", codec=True)
samp = lambda d: SamplingConfig(**{k: d[k] for k in SamplingConfig().__dict__ if k in d})
req = GenerationRequest(text_ids=g["text_ids"], language="english", speaker="speaker1")
out = m.generate([req], samp(GREEDY), 64)
assert np.array_equal(out.codes[0], g["greedy_codes"]), "greedy tokens differ"
req.seed = 42
out = m.generate([req, req], samp(SAMPLED), 64)
assert np.array_equal(out.codes[0], g["sampled_codes"]) and np.array_equal(out.codes[1], g["sampled_codes"]), "sampled tokens differ"
w = m.decode_codes(g["codec_codes"][None, :20])[0]
assert len(w) == len(g["wav_full20"]) and snr_db(g["wav_full20"], w) >= 40.0, "waveform differs"
w = m.decode_codes(g["codec_codes"][None], chunk=-2)[0]
assert len(w) == len(g["wav_policy600"]) and snr_db(g["wav_policy600"], w) >= 40.0, "chunked waveform differs"
m.close()
print("variant ok")
"""


@pytest.mark.parametrize("env", [{"QTTS_CONV_HALO": "0"}, {"QTTS_CONV_TILE": "3"}, {"QTTS_GEMM_KIND": "0"},
                                 {"QTTS_CONV_TILE2_K": "0"}, {"QTTS_CONV_TILE2": "0"}, {"QTTS_CONV_UNIT": "0"}, {"QTTS_ROWS_MIN_M": "17"}, {"QTTS_GEMM_MIN_BLOCKS": "448"},
                                 {"QTTS_ATTN_TILES": "0"}, {"QTTS_CONV_XCD": "0"}],
                         ids=["tap-gemm-conv", "wide-conv-tiles", "in-block-split-k-gemm", "wave-grid-conv-everywhere",
                              "no-wave-grid-conv", "unfused-residual-units", "large-m-gemm-from-17-rows", "more-split-k-slabs",
                              "per-token-prefill-attention", "conv-tiles-in-launch-order"])
def test_variant_matches_golden(env):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT], env=e, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and "variant ok" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


MEMO_SCRIPT = r"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, ROOT)
from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synthetic_model
m, cfg, ccfg, tmpl = synthetic_model("M0", dtype="bf16", codec=False)
rng = np.random.default_rng(5)
reqs = [GenerationRequest(text_ids=rng.integers(10, tmpl["text_hi"], size=6 + 3 * i).astype(np.int32), language="english",
                          speaker="speaker1", seed=100 + i) for i in range(5)]
out = m.generate(reqs, SamplingConfig(), 24)
h = hashlib.sha256()
for c in out.codes:
    h.update(np.ascontiguousarray(c).tobytes())
m.close()
print("codes", h.hexdigest())
"""


def test_tabulated_pass_starts_change_nothing():
    """The code predictor's tabulated pass starts (projection + first-layer q|k|v per (codebook, code)) hold exactly what the
    launches they replace compute: sampled bf16 codes are the same with the tables switched off."""
    digests = []
    for env in ({"QTTS_CP_FRAG": "0"}, {"QTTS_CP_FRAG": "0", "QTTS_CP_MEMO": "0"}):   # (the tables feed the five-launch layers here)
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + MEMO_SCRIPT], env=e, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0 and "codes " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        digests.append(r.stdout.split("codes ")[1].split()[0])
    assert digests[0] == digests[1]


UNIT_SCRIPT = r"""
import sys, os, hashlib
import numpy as np
sys.path.insert(0, ROOT)
from qwen3_tts_rs_amd import synthetic_model
m, cfg, ccfg, tmpl = synthetic_model("M0", dtype="bf16", codec=True, talker=False)
q = ccfg["decoder_config"]["num_quantizers"]
codes = np.random.default_rng(11).integers(0, ccfg["decoder_config"]["codebook_size"], (3, 40, q)).astype(np.uint32)
h = hashlib.sha256()
for w in m.decode_codes(codes, lens=[40, 33, 17]):
    h.update(np.ascontiguousarray(w).tobytes())
m.close()
print("wav", h.hexdigest())
"""


def test_units_reading_the_f32_stream_change_nothing():
    """bf16 vocoder: a fused residual unit that activates the f32 stream itself computes, for every element, the same function of
    the same f32 value as the producer that used to store the activated copy - the waveforms are equal bit for bit."""
    digests = []
    for env in ({}, {"QTTS_UNIT_F32IN": "0"}):
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + UNIT_SCRIPT], env=e, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0 and "wav " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        digests.append(r.stdout.split("wav ")[1].split()[0])
    assert digests[0] == digests[1]


def test_the_f32_stream_epilogue_pass_changes_nothing():
    """The conv kernels' epilogue pass for the vocoder's usual tail (bias, f32 residual, f32 stream and / or SnakeBeta copy) makes
    the same values and the same stores as the general tail it stands in for: the waveforms are equal bit for bit."""
    for dtype in ("bf16", "f32"):
        digests = []
        for env in ({}, {"QTTS_EPI_STREAM": "0"}):
            e = dict(os.environ, **env)
            r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + UNIT_SCRIPT.replace('dtype="bf16"', 'dtype="%s"' % dtype)], env=e,
                               capture_output=True, text=True, timeout=280)
            assert r.returncode == 0 and "wav " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
            digests.append(r.stdout.split("wav ")[1].split()[0])
        assert digests[0] == digests[1], dtype


FINAL_SCRIPT = r"""
import sys, os
import numpy as np
sys.path.insert(0, ROOT)
import torch
from qwen3_tts_rs_amd import Model, LoaderConfig, synth
cfg, ccfg, tmpl = synth.shape_set("S06")
m = Model(cfg, ccfg, LoaderConfig(dtype="bf16"))
for k, t in synth.iter_weights_torch(synth.codec_keys(ccfg), 20261111, "cuda:0", True):
    m.load_tensor(k, t)
m.finalize()
codes = np.random.default_rng(5).integers(0, 2048, (3, 70, 16)).astype(np.uint32)
wavs = m.decode_codes(codes, [70, 41, 9])
m.close()
np.save(sys.argv[1], np.concatenate(wavs))
print("saved", sum(len(w) for w in wavs))
"""


def test_output_conv_inside_the_last_unit_agrees_with_the_separate_kernel(tmp_path):
    """Reference-default vocoder in bf16: the output conv computed inside the last 96-channel unit's kernel (from the activated rows
    in LDS, rounded to bf16 as the stored copy would be) against the separate kernel reading that copy from memory.  The same
    products summed in another order: equal to f32 rounding, ragged strips and the strips' first samples included."""
    out = []
    for tag, env in (("fused", {}), ("separate", {"QTTS_FINAL_FUSED": "0"})):
        path = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FINAL_SCRIPT, path], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=280)
        assert r.returncode == 0 and "saved " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        out.append(np.load(path))
    a, b = out
    assert a.shape == b.shape and np.isfinite(a).all()
    err = float(np.abs(a.astype(np.float64) - b).max())
    assert err <= 2e-6, err   # samples are in [-1, 1]: a few f32 ulps of the 672-term sums


def test_two_units_in_one_launch_change_nothing(tmp_path):
    """Reference-default vocoder in bf16: the first two residual units of the 96-channel block in one kernel (the stream between
    them stays in registers and LDS) against a launch per unit - every value goes through the same arithmetic, so the waveforms
    are equal bit for bit (ragged strips included: rows without output must pass exact zeros on)."""
    out = []
    for tag, env in (("pair", {}), ("single", {"QTTS_UNIT96_PAIR": "0"})):
        path = str(tmp_path / (tag + ".npy"))
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FINAL_SCRIPT, path], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=280)
        assert r.returncode == 0 and "saved " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        out.append(np.load(path))
    assert out[0].shape == out[1].shape and np.array_equal(out[0], out[1])


def test_k_split_row_tiles_change_nothing():
    """Two 16-row token tiles per k_split block (the choice with several decode streams in flight) add the same products in
    the same order as one tile per block: sampled bf16 codes of a 20-row batch are identical."""
    script = MEMO_SCRIPT.replace("for i in range(5)]", "for i in range(20)]")
    assert script != MEMO_SCRIPT
    digests = []
    for rows in ("16", "32"):
        e = dict(os.environ, QTTS_TUNE="shared", QTTS_KS_ROWS=rows)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + script], env=e, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0 and "codes " in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        digests.append(r.stdout.split("codes ")[1].split()[0])
    assert digests[0] == digests[1]


FRAG_SCRIPT = r"""
import sys, os
import numpy as np
sys.path.insert(0, ROOT)
from qwen3_tts_rs_amd import GenerationRequest, SamplingConfig, synthetic_model
m, cfg, ccfg, tmpl = synthetic_model("M0", dtype="bf16", codec=False)
rng = np.random.default_rng(5)
reqs = [GenerationRequest(text_ids=rng.integers(10, tmpl["text_hi"], size=6 + 3 * (i % 7)).astype(np.int32), language="english",
                          speaker="speaker1", seed=100 + i) for i in range(int(os.environ.get("FRAG_ROWS", "5")))]
out = m.generate(reqs, SamplingConfig(), 12, trace=True)
graph = m.generate(reqs, SamplingConfig(), 12)
assert np.array_equal(graph.codes, out.codes), "graph replay differs from kernel-by-kernel launches"
np.savez(sys.argv[1], codes=out.codes, cp_logits=out.cp_logits, logits=out.logits)
m.close()
print("saved")
"""


def test_fragment_order_code_predictor_passes_agree_with_the_five_launch_layers(tmp_path):
    """The single-token code-predictor passes through the fragment-order GEMM (the default where its shapes fit) against the same
    passes through the LDS-tile kernels: the two sum K in different orders, so the code-predictor logits agree to bf16 rounding
    for as long as the sampled histories agree (five live rows of the 64 the kernels compute)."""
    import numpy as np
    res = []
    for i, env in enumerate(({}, {"QTTS_CP_FRAG": "0"})):
        path = str(tmp_path / f"r{i}.npz")
        e = dict(os.environ, **env)
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FRAG_SCRIPT, path], env=e, capture_output=True, text=True, timeout=280)
        assert r.returncode == 0 and "saved" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        res.append(np.load(path))
    a, b = res
    checked = 0
    for row in range(a["codes"].shape[0]):
        same = (a["codes"][row] == b["codes"][row]).all(axis=1)
        upto = int(np.argmin(same)) if not same.all() else len(same)
        for t in range(min(upto + 1, len(same))):
            agree = a["codes"][row][t][1:] == b["codes"][row][t][1:]
            if t > 0 and not (a["codes"][row][t - 1] == b["codes"][row][t - 1]).all():
                break
            if a["codes"][row][t][0] != b["codes"][row][t][0]:
                break
            nj = min((int(np.argmin(agree)) if not agree.all() else len(agree)) + 1, len(agree))
            err = np.abs(a["cp_logits"][row][t][:nj] - b["cp_logits"][row][t][:nj]).max()
            assert err <= 2e-2, f"row {row} frame {t}: {err:.4f}"
            checked += nj
    assert checked >= 5 * 15     # at least the first frame of every row


def _cp_logits_agree(a, b, rows_needed):
    import numpy as np
    checked = 0
    for row in range(a["codes"].shape[0]):
        for t in range(a["codes"].shape[1]):
            if t > 0 and not (a["codes"][row][t - 1] == b["codes"][row][t - 1]).all():
                break
            if a["codes"][row][t][0] != b["codes"][row][t][0]:
                break
            agree = a["codes"][row][t][1:] == b["codes"][row][t][1:]
            nj = min((int(np.argmin(agree)) if not agree.all() else len(agree)) + 1, len(agree))
            err = np.abs(a["cp_logits"][row][t][:nj] - b["cp_logits"][row][t][:nj]).max()
            assert err <= 2e-2, f"row {row} frame {t}: {err:.4f}"
            checked += nj
    assert checked >= rows_needed * 15


def test_one_wave_short_context_attention_agrees_with_the_general_kernel(tmp_path):
    """With several decode streams in flight the code predictor's passes (at most 16 keys) run their attention with one wave per
    (token, kv head) instead of a block of four: same arithmetic up to the softmax's summation order, so the code-predictor logits
    agree to bf16 rounding for as long as the sampled histories agree.  Both runs use the kernel selection of shared mode."""
    import numpy as np
    res = []
    for i, env in enumerate(({"QTTS_TUNE": "shared"}, {"QTTS_TUNE": "shared", "QTTS_ATTN_SMALL": "0"})):
        path = str(tmp_path / f"s{i}.npz")
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FRAG_SCRIPT, path], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=280)
        assert r.returncode == 0 and "saved" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        res.append(np.load(path))
    assert not np.array_equal(res[0]["cp_logits"], res[1]["cp_logits"]), "the switch selected the same kernel twice"
    _cp_logits_agree(res[0], res[1], 5)


def test_eight_waves_per_attention_block_agree_with_four(tmp_path):
    """A stream that has the GPU to itself walks the talker's growing contexts with eight waves per attention block instead of
    four: the same online softmax over twice the partial results, so the talker's logits agree to bf16 rounding for as long as
    the sampled histories agree."""
    import numpy as np
    res = []
    for i, env in enumerate(({"QTTS_TUNE": "alone"}, {"QTTS_TUNE": "alone", "QTTS_ATTN_WAVES": "4"})):
        path = str(tmp_path / f"a{i}.npz")
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FRAG_SCRIPT, path], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=280)
        assert r.returncode == 0 and "saved" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        res.append(np.load(path))
    a, b = res
    assert not np.array_equal(a["logits"], b["logits"]), "the switch selected the same kernel twice"
    checked = 0
    for row in range(a["codes"].shape[0]):
        for t in range(a["codes"].shape[1]):
            if t > 0 and not (a["codes"][row][t - 1] == b["codes"][row][t - 1]).all():
                break
            err = np.abs(a["logits"][row][t] - b["logits"][row][t]).max()
            assert err <= 2e-2, f"row {row} frame {t}: {err:.4f}"
            checked += 1
    assert checked >= 2 * a["codes"].shape[0]


def test_fragment_order_passes_at_more_than_64_rows(tmp_path):
    """70 rows: the fragment-order GEMMs then run on 128 packed rows (eight row tiles instead of four); same check as at 5 rows."""
    import numpy as np
    res = []
    for i, env in enumerate(({"FRAG_ROWS": "70"}, {"FRAG_ROWS": "70", "QTTS_CP_FRAG": "0"})):
        path = str(tmp_path / f"w{i}.npz")
        r = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + FRAG_SCRIPT, path], env=dict(os.environ, **env), capture_output=True,
                           text=True, timeout=280)
        assert r.returncode == 0 and "saved" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])
        res.append(np.load(path))
    assert not np.array_equal(res[0]["cp_logits"], res[1]["cp_logits"]), "the switch selected the same kernels twice"
    _cp_logits_agree(res[0], res[1], 70)
