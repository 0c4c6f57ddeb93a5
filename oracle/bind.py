"""ctypes binding of the CPU oracle (TEST INFRASTRUCTURE: used by tests/, smoke() and bench.py's
cpu_baseline leg only -- never by the shipped path)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build(force: bool = False):
    out = os.path.join(_HERE, "_build", "liboracle.so")
    if force or not os.path.exists(out):
        subprocess.check_call(["make", "-C", _HERE, "-j4"], stdout=subprocess.DEVNULL)
    return out


def _load(fast: bool):
    build()
    path = os.path.join(_HERE, "_build", "liboracle_fast.so" if fast else "liboracle.so")
    lib = C.CDLL(path)
    lib.orc_create.restype = C.c_void_p
    lib.orc_last_error.restype = C.c_char_p
    lib.orc_sample_token.restype = C.c_int64
    return lib


_LIBS = {}


def lib(fast: bool = False):
    if fast not in _LIBS:
        _LIBS[fast] = _load(fast)
    return _LIBS[fast]


def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _attn(t):
    sw = t.get("sliding_window_effective", -1)
    return np.array([t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"], t["num_attention_heads"],
                     t["num_key_value_heads"], t["head_dim"], t.get("rms_norm_eps", 1e-6), t.get("rope_theta", 1e4),
                     sw if sw else -1, 1.0 if t.get("attention_bias") else 0.0], np.float64)


def sampling_array(s: dict) -> np.ndarray:
    """temperature, top_k, top_p, do_sample, repetition_penalty, min_new_tokens, eos, suppress_lo, suppress_hi,
    subtalker_do_sample, subtalker_temperature, subtalker_top_k, subtalker_top_p"""
    return np.array([s.get("temperature", 0.9), s.get("top_k", 50), s.get("top_p", 1.0),
                     1.0 if s.get("do_sample", True) else 0.0, s.get("repetition_penalty", 1.05),
                     s.get("min_new_tokens", 2), s.get("eos_token_id", -1), s.get("suppress_lo", 0),
                     s.get("suppress_hi", 0), 1.0 if s.get("subtalker_do_sample", True) else 0.0,
                     s.get("subtalker_temperature", 0.9), s.get("subtalker_top_k", 50),
                     s.get("subtalker_top_p", 1.0)], np.float64)


class Oracle:
    def __init__(self, fast: bool = False):
        self.lib = lib(fast)
        self.h = C.c_void_p(self.lib.orc_create())
        self.cfg = None
        self.codec_cfg = None

    def __del__(self):
        try:
            self.lib.orc_free(self.h)
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.orc_last_error().decode())

    def load_tensor(self, key: str, arr: np.ndarray):
        a = np.ascontiguousarray(arr, np.float32)
        shape = np.array(a.shape, np.int64)
        self._chk(self.lib.orc_load_tensor(self.h, key.encode(), _vp(a), _vp(shape), C.c_int(a.ndim)))

    def set_model_config(self, cfg: dict):
        self.cfg = cfg
        t = dict(cfg["talker_config"])
        # the talker hands config.sliding_window to attention unconditionally (config/talker_config.rs:360-362)
        t["sliding_window_effective"] = t.get("sliding_window") or -1
        c = dict(t["code_predictor_config"])
        lt = c.get("layer_types")
        c["sliding_window_effective"] = -1  # default layer types are all full_attention (code_predictor_config.rs:172-186)
        sliding = bool(lt and any(x == "sliding_attention" for x in lt)) or (
            bool(c.get("use_sliding_window", False)) and int(c.get("max_window_layers", 28)) < int(c.get("num_hidden_layers", 5)))
        win = c.get("sliding_window", 4096)
        # the code predictor's context never exceeds num_code_groups + 1 positions: a window at least that long cuts nothing
        # (nn/attention/standard.rs:28-39 with the default window of code_predictor_config.rs:135-137)
        if sliding and win and int(win) < int(t["num_code_groups"]) + 1:
            raise NotImplementedError("per-layer sliding code-predictor attention with a window shorter than its context")
        misc = np.array([t["vocab_size"], t["text_hidden_size"], t["text_vocab_size"], t["num_code_groups"],
                         c["vocab_size"], t["codec_eos_token_id"], t["codec_bos_id"], t["codec_pad_id"],
                         t["codec_think_id"], t["codec_nothink_id"], t["codec_think_bos_id"], t["codec_think_eos_id"],
                         cfg["tts_bos_token_id"], cfg["tts_eos_token_id"], cfg["tts_pad_token_id"]], np.float64)
        ta, ca = _attn(t), _attn(c)
        self._chk(self.lib.orc_set_model_config(self.h, _vp(ta), _vp(ca), _vp(misc)))
        s = cfg.get("speaker_encoder_config")
        if s:
            ch = np.array(s["enc_channels"], np.int32)
            ke = np.array(s["enc_kernel_sizes"], np.int32)
            di = np.array(s["enc_dilations"], np.int32)
            self._chk(self.lib.orc_set_speaker_config(self.h, s["mel_dim"], s["enc_dim"], _vp(ch), _vp(ke), _vp(di),
                                                      len(ch), s["enc_attention_channels"], s["enc_res2net_scale"],
                                                      s["enc_se_channels"]))

    def set_codec_config(self, codec_cfg: dict):
        self.codec_cfg = codec_cfg
        c = codec_cfg["decoder_config"]
        a = np.array([c["codebook_size"], c["hidden_size"], c["latent_dim"], c["codebook_dim"], c["rope_theta"],
                      c["num_attention_heads"], c["num_key_value_heads"], c.get("head_dim") or 0,
                      1.0 if c.get("attention_bias") else 0.0, c["sliding_window"], c["intermediate_size"],
                      c["rms_norm_eps"], c["num_hidden_layers"], c["num_quantizers"]], np.float64)
        r = np.array(c["upsample_rates"], np.int32)
        u = np.array(c["upsampling_ratios"], np.int32)
        self._chk(self.lib.orc_set_codec_config(self.h, _vp(a), c["decoder_dim"], _vp(r), len(r), _vp(u), len(u)))

    # ------------------------------------------------------------------ requests
    def _req(self, text_ids, prefill, speaker_token=-1, speaker_embed=None, instruct_ids=None, non_streaming=True,
             ref_text_ids=None, ref_codes=None, max_new_tokens=0, trace=False, rng_stride=1, rng_offset=0):
        text_ids = np.ascontiguousarray(text_ids, np.uint32)
        prefill = np.ascontiguousarray(prefill, np.int32)
        ins = np.ascontiguousarray(instruct_ids, np.uint32) if instruct_ids is not None else np.zeros(0, np.uint32)
        rt = np.ascontiguousarray(ref_text_ids, np.uint32) if ref_text_ids is not None else np.zeros(0, np.uint32)
        rc = np.ascontiguousarray(ref_codes, np.uint32) if ref_codes is not None else np.zeros((0, 1), np.uint32)
        se = np.ascontiguousarray(speaker_embed, np.float32) if speaker_embed is not None else np.zeros(1, np.float32)
        ints = np.array([len(text_ids), len(ins), len(prefill), speaker_token, 1 if speaker_embed is not None else 0,
                         1 if non_streaming else 0, len(rt), rc.shape[0] if ref_codes is not None else 0,
                         max_new_tokens, 1, 1 if trace else 0, rng_stride, rng_offset], np.int32)
        return text_ids, ins, prefill, se, rt, rc, ints

    def generate(self, text_ids, prefill, sampling: dict, max_new_tokens: int, seed: int = 0, trace: bool = False, **kw):
        t = self.cfg["talker_config"]
        G, V, H = t["num_code_groups"], t["vocab_size"], t["hidden_size"]
        text_ids, ins, prefill, se, rt, rc, ints = self._req(text_ids, prefill, max_new_tokens=max_new_tokens,
                                                            trace=trace, **kw)
        codes = np.zeros((max_new_tokens + 1, G), np.uint32)
        steps, eff = C.c_int(0), C.c_int(0)
        lt = np.zeros((max_new_tokens + 1, V), np.float32) if trace else None
        ht = np.zeros((max_new_tokens + 1, H), np.float32) if trace else None
        mm = C.c_float(0)
        s = sampling_array(sampling)
        self._chk(self.lib.orc_generate(self.h, _vp(text_ids), _vp(ins), _vp(prefill), _vp(se), _vp(rt), _vp(rc),
                                        _vp(ints), _vp(s), C.c_uint64(seed), _vp(codes), C.byref(steps), C.byref(eff),
                                        _vp(lt), _vp(ht), C.byref(mm)))
        n = steps.value
        out = dict(codes=codes[:n].copy(), num_steps=n, effective_len=eff.value, min_margin=mm.value)
        if trace:
            out["logits"] = lt[:n].copy()
            out["hidden"] = ht[:n].copy()
        return out

    def build_prompt(self, text_ids, prefill, **kw):
        H = self.cfg["talker_config"]["hidden_size"]
        text_ids, ins, prefill, se, rt, rc, ints = self._req(text_ids, prefill, **kw)
        cap = len(text_ids) + len(ins) + len(rt) + rc.shape[0] + 32
        prompt = np.zeros((cap, H), np.float32)
        trailing = np.zeros((cap, H), np.float32)
        P, Tt = C.c_int(0), C.c_int(0)
        self._chk(self.lib.orc_build_prompt(self.h, _vp(text_ids), _vp(ins), _vp(prefill), _vp(se), _vp(rt), _vp(rc),
                                            _vp(ints), _vp(prompt), cap, C.byref(P), _vp(trailing), cap, C.byref(Tt)))
        return prompt[:P.value].copy(), trailing[:Tt.value].copy()

    def talker_forward(self, embeds: np.ndarray):
        t = self.cfg["talker_config"]
        e = np.ascontiguousarray(embeds, np.float32)
        T = e.shape[0]
        hid = np.zeros((T, t["hidden_size"]), np.float32)
        log = np.zeros((T, t["vocab_size"]), np.float32)
        self._chk(self.lib.orc_talker_forward(self.h, _vp(e), T, _vp(hid), _vp(log)))
        return hid, log

    def code_predictor(self, hidden: np.ndarray, cb0: int, sampling: dict, seed: int = 0):
        t = self.cfg["talker_config"]
        G, Vc = t["num_code_groups"], t["code_predictor_config"]["vocab_size"]
        h = np.ascontiguousarray(hidden, np.float32)
        codes = np.zeros(G - 1, np.uint32)
        logits = np.zeros((G - 1, Vc), np.float32)
        s = sampling_array(sampling)
        self._chk(self.lib.orc_code_predictor(self.h, _vp(h), int(cb0), _vp(s), C.c_uint64(seed), _vp(codes), _vp(logits)))
        return codes, logits

    def _total_upsample(self):
        c = self.codec_cfg["decoder_config"]
        return int(np.prod(c["upsample_rates"]) * np.prod(c["upsampling_ratios"]))

    def codec_decode(self, codes: np.ndarray, mode: int = 0, chunk: int = 256, left_ctx: int = 32):
        c = np.ascontiguousarray(codes, np.uint32)
        T = c.shape[0]
        cap = (T + 2) * self._total_upsample()
        wav = np.zeros(cap, np.float32)
        n = C.c_int64(0)
        self._chk(self.lib.orc_codec_decode(self.h, _vp(c), T, mode, chunk, left_ctx, _vp(wav), C.c_int64(cap), C.byref(n)))
        return wav[:n.value].copy()

    def codec_trace(self, codes: np.ndarray, stage: int):
        c = np.ascontiguousarray(codes, np.uint32)
        T = c.shape[0]
        d = self.codec_cfg["decoder_config"]
        cap = (T + 2) * self._total_upsample() * max(d["decoder_dim"], d["latent_dim"])
        out = np.zeros(cap, np.float32)
        n = C.c_int64(0)
        self._chk(self.lib.orc_codec_trace(self.h, _vp(c), T, stage, _vp(out), C.c_int64(cap), C.byref(n)))
        return out[:n.value].copy()

    def encode_audio(self, audio: np.ndarray, valid_q: int = 16):
        a = np.ascontiguousarray(audio, np.float32)
        cap = (len(a) // 1920 + 2) * valid_q
        codes = np.zeros(cap, np.uint32)
        T = C.c_int(0)
        self._chk(self.lib.orc_encode_audio(self.h, _vp(a), len(a), valid_q, _vp(codes), C.c_int64(cap), C.byref(T)))
        return codes[:T.value * valid_q].reshape(T.value, valid_q).copy()

    def encode_audio_margins(self, audio: np.ndarray, valid_q: int = 16):
        """(codes [T][q], gap [T][q], scale [T][q]): gap = second nearest - nearest codebook distance, scale = magnitude of the
        nearest distance's two terms; gap / scale at float32 rounding level marks a near-tie."""
        a = np.ascontiguousarray(audio, np.float32)
        cap = (len(a) // 1920 + 2) * valid_q
        codes = np.zeros(cap, np.uint32)
        mg = np.zeros(cap * 2, np.float32)
        T = C.c_int(0)
        self._chk(self.lib.orc_encode_audio_margins(self.h, _vp(a), len(a), valid_q, _vp(codes), _vp(mg), C.c_int64(cap), C.byref(T)))
        n = T.value * valid_q
        mg = mg[:2 * n].reshape(T.value, valid_q, 2)
        return codes[:n].reshape(T.value, valid_q).copy(), mg[:, :, 0].copy(), mg[:, :, 1].copy()

    def speaker_embed(self, mel: np.ndarray):
        m = np.ascontiguousarray(mel, np.float32)
        out = np.zeros(self.cfg["speaker_encoder_config"]["enc_dim"], np.float32)
        self._chk(self.lib.orc_speaker_embed(self.h, _vp(m), m.shape[0], _vp(out)))
        return out

    def mel(self, audio: np.ndarray):
        a = np.ascontiguousarray(audio, np.float32)
        cap = (len(a) // 256 + 8) * 128
        out = np.zeros(cap, np.float32)
        nf = C.c_int(0)
        self._chk(self.lib.orc_mel(_vp(a), len(a), _vp(out), C.c_int64(cap), C.byref(nf)))
        return out[:nf.value * 128].reshape(nf.value, 128).copy()


def resolve_codec_prefill(cfg: dict, language: str, speaker: str | None):
    """build_codec_prefill (qwen_tts/src/nn/generation.rs:452-498) -- pure id bookkeeping."""
    t = cfg["talker_config"]
    lang = language.lower()
    if lang in ("chinese", "auto") and speaker and t.get("spk_is_dialect"):
        dv = t["spk_is_dialect"].get(speaker.lower())
        if isinstance(dv, str):
            lang = dv.lower()
    auto = [t["codec_nothink_id"], t["codec_think_bos_id"], t["codec_think_eos_id"]]
    if lang == "auto":
        return auto
    lm = t.get("codec_language_id") or {}
    if lang in lm:
        return [t["codec_think_id"], t["codec_think_bos_id"], lm[lang], t["codec_think_eos_id"]]
    return auto


def resolve_speaker_token(cfg: dict, speaker: str | None) -> int:
    if not speaker:
        return -1
    m = cfg["talker_config"].get("spk_id") or {}
    return m.get(speaker.lower(), -1)
