"""Shared fixtures.  `-m "not gpu"` tests run on a CPU-only machine; `-m gpu` tests call the HIP library through
its C ABI and compare with the CPU oracle (oracle/), which is test infrastructure only."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SEED = 20261111
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (runs the HIP library)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are skipped (not failed) on machines without a device unless they were asked for with -m gpu
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU in this machine")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def bind():
    from oracle import bind as b
    b.build()
    return b


def make_oracle(bind, shape="T0", dtype="f32", model_type="custom_voice", talker=True, codec=False, fast=False):
    """Oracle loaded with the synthetic weights of a shape set (bf16 runs share bf16-rounded weights)."""
    from qwen3_tts_rs_amd import synth, round_bf16
    cfg, ccfg, tmpl = synth.shape_set(shape, model_type)
    o = bind.Oracle(fast=fast)
    keys = []
    if talker:
        keys += synth.talker_keys(cfg) + synth.speaker_keys(cfg)
    if codec:
        keys += synth.codec_keys(ccfg)
    for k, a in synth.iter_weights(keys, SEED):
        o.load_tensor(k, round_bf16(a) if dtype == "bf16" else a)
    if talker:
        o.set_model_config(cfg)
    if codec:
        o.set_codec_config(ccfg)
    return o, cfg, ccfg, tmpl


@pytest.fixture(scope="session")
def oracle_t0(bind):
    return make_oracle(bind, "T0", "f32", talker=True, codec=True)


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(GOLDEN, "t0_golden.npz")
    return np.load(path, allow_pickle=False)


def snr_db(ref, x):
    n = min(len(ref), len(x))
    e = np.asarray(ref[:n], np.float64) - np.asarray(x[:n], np.float64)
    return 10 * np.log10(max((np.asarray(ref[:n], np.float64) ** 2).sum(), 1e-30) / max((e ** 2).sum(), 1e-30))
