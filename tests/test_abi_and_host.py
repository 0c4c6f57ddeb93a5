"""CPU-only checks of the drop-in boundary and the host-side mirror: the C-ABI library loads and exports every
symbol include/qtts.h declares (no compute calls without a GPU), option merging follows model/config.rs, the
synthetic weight factory is deterministic."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, SEED


def header_symbols():
    hdr = open(os.path.join(ROOT, "include", "qtts.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(qtts_[a-z_0-9]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from qwen3_tts_rs_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the HIP library first (__graft_entry__.build())"
    lib = C.CDLL(_lib.LIB_PATH)
    names = header_symbols()
    assert len(names) >= 16
    for n in names:
        assert hasattr(lib, n), f"{n} is declared in include/qtts.h but not exported"
    assert set(_lib.EXPORTS) <= set(names)


def test_ctypes_structs_match_the_header_layout(tmp_path):
    """Size and every field offset of the ctypes mirrors equal what a C compiler lays out from include/qtts.h."""
    import subprocess
    from qwen3_tts_rs_amd import _lib
    structs = {"qtts_request": _lib.Request, "qtts_sampling": _lib.Sampling, "qtts_codes_out": _lib.CodesOut,
               "qtts_gen_debug": _lib.GenDebug}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "qtts.h"', 'int main(void) {']
    for cname, st in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in st._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = {}
    for ln in subprocess.check_output([str(exe)], text=True).split("\n"):
        if ln:
            c, f, v = ln.split()
            got[(c, f)] = int(v)
    for cname, st in structs.items():
        assert got[(cname, "size")] == C.sizeof(st), cname
        for fname, _ in st._fields_:
            assert got[(cname, fname)] == getattr(st, fname).offset, f"{cname}.{fname}"
    # every field of the header's structs is mirrored (a field added to the header alone would shift nothing but be lost)
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qtts.h")).read(), flags=re.S)
    for cname, st in structs.items():
        body = re.search(r"typedef struct \{([^}]*)\}\s*" + cname + ";", hdr).group(1)
        n_fields = len([d for d in body.split(";") if d.strip()])
        assert n_fields == len(st._fields_), cname


def test_header_cites_the_reference_for_every_entry_point():
    hdr = open(os.path.join(ROOT, "include", "qtts.h")).read()
    assert hdr.count(".rs:") >= 10
    assert 'extern "C"' in hdr
    code = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    assert "torch" not in code.lower() and "std::" not in code      # plain C types only in the signatures


def test_error_reporting_without_device():
    """Argument errors are reported through return codes + qtts_last_error, never by unwinding."""
    from qwen3_tts_rs_amd import _lib
    lib = _lib.load()
    rc = lib.qtts_model_finalize(None)
    assert rc != 0
    assert b"null" in lib.qtts_last_error().lower() or len(lib.qtts_last_error()) > 0
    out = (C.c_double * 8)()
    assert lib.qtts_last_timing(None, out, 8) != 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from qwen3_tts_rs_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libqtts_hip.so"))
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.load()


def test_product_path_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "qwen3-tts-rs_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in txt.lower(), f"{f} mentions the oracle"


def test_generate_config_merge():
    from qwen3_tts_rs_amd import GenerateConfig, SamplingConfig
    d = GenerateConfig.from_json(json.dumps(dict(do_sample=True, top_k=40, temperature=0.8, subtalker_dosample=False,
                                                 max_new_tokens=1000)))
    assert d.top_k == 40 and d.subtalker_do_sample is False and d.max_new_tokens == 1000
    s = d.merge(top_k=None, temperature=0.5)
    assert s.top_k == 40 and s.temperature == 0.5 and s.repetition_penalty == 1.05
    assert d.effective_max_tokens(None) == 1000 and d.effective_max_tokens(7) == 7
    g = SamplingConfig.cli_greedy()
    # --greedy keeps the repetition penalty and top-k defaults (synthesis/synthesize_voice.rs:30-38)
    assert (g.do_sample, g.subtalker_do_sample, g.temperature, g.repetition_penalty, g.top_k) == (False, False, 1.0, 1.05, 50)


def test_synthetic_weights_are_deterministic():
    from qwen3_tts_rs_amd import synth
    a = synth.make_tensor("talker.model.layers.0.mlp.down_proj.weight", (8, 16), "normal:0.02", SEED)
    b = synth.make_tensor("talker.model.layers.0.mlp.down_proj.weight", (8, 16), "normal:0.02", SEED)
    c = synth.make_tensor("talker.model.layers.1.mlp.down_proj.weight", (8, 16), "normal:0.02", SEED)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert abs(float(a.std()) - 0.02) < 0.01
    # value pinned so that a silent change of the generator invalidates the golden fixtures loudly
    assert a.dtype == np.float32 and a.shape == (8, 16)
    cfg, ccfg, tmpl = synth.shape_set("S17")
    w = synth.weight_bytes_per_frame(cfg, 2)
    assert abs(w["total"] / 1e9 - 5.316) < 0.01     # algorithmic weight bytes per frame of the 1.7B-like shape set


def test_text_template():
    from qwen3_tts_rs_amd import synth
    _, _, tmpl = synth.shape_set("T0")
    ids = synth.synthetic_text_ids(tmpl, 12, 0)
    assert len(ids) == 12 + 8 and ids[:3].tolist() == tmpl["prefix"] and ids[-5:].tolist() == tmpl["suffix"]


def test_lpt_assignment_balances_and_is_deterministic():
    from qwen3_tts_rs_amd.shard import assign_utterances, gather_results
    costs = [5, 9, 1, 7, 3, 3, 8, 2]
    a = assign_utterances(costs, 3)
    assert sorted(i for r in a for i in r) == list(range(8))
    loads = [sum(costs[i] for i in r) for r in a]
    assert max(loads) - min(loads) <= max(costs)
    assert a == assign_utterances(costs, 3)
    assert assign_utterances(costs, 1) == [list(range(8))]
    assert assign_utterances([], 2) == [[], []]
    out = gather_results([2, 0], ["c", "a"], 3)
    assert out == ["a", None, "c"]


def test_strong_scaling_job_is_dealt_exactly_once():
    """bench.py --mode strong: every utterance of the fixed job goes to exactly one rank, shares are balanced."""
    import types
    import bench
    args = types.SimpleNamespace(utterances=256)
    for world in (1, 2, 4, 8):
        shares = [bench.strong_share(args, r, world) for r in range(world)]
        assert sorted(u for s in shares for u in s) == list(range(256))
        assert max(len(s) for s in shares) - min(len(s) for s in shares) <= 1
    clip = bench.reference_clip(5)
    assert clip.dtype == np.float32 and 72000 <= len(clip) <= 72000 + 641 * 256 and np.abs(clip).max() < 1.0


def test_rust_sys_crate_declares_the_whole_header():
    """qwen-tts-hip-sys/src/lib.rs: every function of include/qtts.h, every struct field in the header's order."""
    rs = open(os.path.join(ROOT, "qwen-tts-hip-sys", "src", "lib.rs")).read()
    for f in ("Cargo.toml", "build.rs"):
        assert os.path.exists(os.path.join(ROOT, "qwen-tts-hip-sys", f))
    declared = set(re.findall(r"pub fn (qtts_[a-z_0-9]+)\s*\(", rs))
    assert declared == set(header_symbols())
    hdr = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "qtts.h")).read(), flags=re.S)
    ctype = {"int": "c_int", "double": "c_double", "uint64_t": "u64", "const char*": "*const c_char", "const uint32_t*": "*const u32",
             "const float*": "*const c_float", "uint32_t*": "*mut u32", "float*": "*mut c_float", "void*": "*mut c_void"}
    for cname in ("qtts_request", "qtts_sampling", "qtts_codes_out", "qtts_gen_debug"):
        body = re.search(r"typedef struct \{([^}]*)\}\s*" + cname + ";", hdr).group(1)
        c_fields = []
        for d in body.split(";"):
            d = " ".join(d.split())
            if not d:
                continue
            if "(*" in d:                                   # the callback pointer
                c_fields.append((re.search(r"\(\*(\w+)\)", d).group(1), None))
                continue
            m = re.match(r"(.*?)\s*(\w+)$", d)
            c_fields.append((m.group(2), m.group(1).replace(" *", "*")))
        rbody = re.search(r"pub struct " + cname + r" \{(.*?)\n\}", rs, flags=re.S).group(1)
        r_fields = re.findall(r"pub (\w+): ([^,\n]+),", rbody)
        assert [f for f, _ in r_fields] == [f for f, _ in c_fields], cname
        for (f, rt), (_, ct) in zip(r_fields, c_fields):
            if ct is not None:
                assert rt.strip() == ctype[ct], f"{cname}.{f}: {rt} vs {ct}"
    feat = open(os.path.join(ROOT, "qwen-tts-hip-sys", "qwen_tts_hip_feature", "hip.rs")).read()
    for fn in ("qtts_model_create", "qtts_model_load_tensor", "qtts_model_finalize", "qtts_generate", "qtts_codec_decode", "qtts_encode_audio",
               "qtts_mel", "qtts_speaker_embed", "qtts_speaker_embed_mel", "qtts_session_open", "qtts_codec_stream_push"):
        assert "sys::" + fn in feat, fn


def test_host_example_compiles_against_the_header_alone(tmp_path):
    """tools/qtts_synth.cpp is host code on include/qtts.h and nothing else: it compiles with plain g++ (no HIP headers), and the
    header itself is valid C."""
    import subprocess
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tools", "qtts_synth.cpp")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    probe = tmp_path / "probe.c"
    probe.write_text('#include "qtts.h"\nint main(void) { return QTTS_ERR_IO == -6 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(probe)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
