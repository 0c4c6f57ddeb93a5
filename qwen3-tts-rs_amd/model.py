"""Host-side mirror of the reference's public model API over the libqtts_hip C ABI.

Names, argument meaning and error behaviour follow the `qwen_tts` crate:
ModelLoader / LoaderConfig (qwen_tts/src/model/loader.rs:31-54,109-118,267), Model and its
generate_* methods (model/generate.rs:227,343,452,809), GenerateConfig::merge (model/config.rs:128-157),
GenerationResult (model/types.rs:53-69), decode_output (model/generate.rs:29-143).
All arithmetic happens in the HIP library; this module only marshals buffers.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from dataclasses import dataclass, field
from typing import Iterable, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import QTTS_BF16, QTTS_F32, QttsError, check


@dataclass
class GenerateConfig:  # model/config.rs:9-47
    do_sample: bool = True
    top_k: int = 50
    top_p: float = 1.0
    temperature: float = 0.9
    repetition_penalty: float = 1.05
    subtalker_do_sample: bool = True
    subtalker_top_k: int = 50
    subtalker_top_p: float = 1.0
    subtalker_temperature: float = 0.9
    max_new_tokens: int = 2048

    @classmethod
    def from_json(cls, text: str) -> "GenerateConfig":  # model/config.rs:51-87 (note the key `subtalker_dosample`)
        raw = json.loads(text)
        d = cls()
        for k in ("do_sample", "top_k", "top_p", "temperature", "repetition_penalty", "subtalker_top_k",
                  "subtalker_top_p", "subtalker_temperature", "max_new_tokens"):
            if raw.get(k) is not None:
                setattr(d, k, raw[k])
        if raw.get("subtalker_dosample") is not None:
            d.subtalker_do_sample = raw["subtalker_dosample"]
        return d

    def merge(self, **user) -> "SamplingConfig":  # model/config.rs:128-157
        def pick(k):
            v = user.get(k)
            return getattr(self, k) if v is None else v
        return SamplingConfig(do_sample=pick("do_sample"), top_k=pick("top_k"), top_p=pick("top_p"),
                              temperature=pick("temperature"), repetition_penalty=pick("repetition_penalty"),
                              subtalker_do_sample=pick("subtalker_do_sample"), subtalker_top_k=pick("subtalker_top_k"),
                              subtalker_top_p=pick("subtalker_top_p"),
                              subtalker_temperature=pick("subtalker_temperature"))

    def effective_max_tokens(self, user_max: Optional[int]) -> int:
        return self.max_new_tokens if user_max is None else user_max


@dataclass
class SamplingConfig:  # nn/sampling.rs:13-64 (suppress list / eos / min_new_tokens are set inside generate)
    temperature: float = 0.9
    top_k: int = 50
    top_p: float = 1.0
    do_sample: bool = True
    repetition_penalty: float = 1.05
    subtalker_do_sample: bool = True
    subtalker_temperature: float = 0.9
    subtalker_top_k: int = 50
    subtalker_top_p: float = 1.0
    min_new_tokens: int = 2      # nn/sampling.rs:28-29,55

    @classmethod
    def cli_greedy(cls) -> "SamplingConfig":
        """`qwen-tts --greedy`: do_sample=false for both stages, temperature 1.0; repetition penalty and top-k keep
        their defaults (synthesis/synthesize_voice.rs:30-38)."""
        return cls(do_sample=False, subtalker_do_sample=False, temperature=1.0)

    def as_dict(self):
        return dict(self.__dict__)

    def to_c(self) -> _lib.Sampling:
        return _lib.Sampling(self.temperature, self.top_k, self.top_p, int(self.do_sample), self.repetition_penalty,
                             int(self.subtalker_do_sample), self.subtalker_temperature, self.subtalker_top_k,
                             self.subtalker_top_p, int(self.min_new_tokens))


@dataclass
class GenerationRequest:  # GenerationOptions (nn/generation_options.rs) + text ids
    text_ids: Sequence[int]
    language: str = "auto"
    speaker: Optional[str] = None
    instruct_ids: Optional[Sequence[int]] = None
    speaker_embed: Optional[np.ndarray] = None
    ref_codes: Optional[np.ndarray] = None
    ref_text_ids: Optional[Sequence[int]] = None
    non_streaming_mode: bool = True
    seed: Optional[int] = None
    max_new_tokens: int = 0      # continuous batching only: this utterance's own frame limit (0 = the call's)


@dataclass
class Output:  # nn/generation.rs:31-38
    codes: np.ndarray            # (batch, num_steps, groups) u32
    num_steps: int
    effective_lengths: list


@dataclass
class GenerationResult:  # model/types.rs:53-69
    audio: list                  # per-sample float32 waveforms
    sample_rate: int
    codes: Optional[np.ndarray] = None
    effective_lengths: Optional[list] = None


@dataclass
class VoiceClonePromptItem:  # model/voice_clone.rs:64-181 (the transcript arrives tokenised: tokenize_ref_text output)
    ref_spk_embedding: np.ndarray
    ref_code: Optional[np.ndarray] = None          # (codes_len, num_quantizers)
    x_vector_only_mode: bool = True
    ref_text_ids: Optional[Sequence[int]] = None   # ids of `<|im_start|>assistant\n{ref_text}<|im_end|>\n`

    @classmethod
    def x_vector_only(cls, ref_spk_embedding):
        return cls(ref_spk_embedding=ref_spk_embedding)

    @classmethod
    def icl(cls, ref_code, ref_spk_embedding, ref_text_ids):
        return cls(ref_spk_embedding=ref_spk_embedding, ref_code=ref_code, x_vector_only_mode=False, ref_text_ids=ref_text_ids)

    def is_icl(self) -> bool:
        return not self.x_vector_only_mode

    def validate(self):
        if self.is_icl():
            if self.ref_code is None:
                raise QttsError(-5, "ICL mode requires ref_code to be present")
            if self.ref_text_ids is None:
                raise QttsError(-5, "ICL mode requires ref_text to be present")


@dataclass
class LoaderConfig:  # model/loader.rs:31-54
    dtype: str = "bf16"
    load_tokenizer: bool = True
    load_text_tokenizer: bool = True
    device: int = 0
    load_encoder: bool = True   # codec encoder tensors (`encoder.*`): needed only for encode_audio / ICL voice clone


class Model:
    """ConditionalGeneration + TokenizerV2 behind one handle (model/mod.rs:128-140)."""

    def __init__(self, config: dict, codec_config: Optional[dict], loader: LoaderConfig = LoaderConfig(),
                 generate_defaults: Optional[GenerateConfig] = None):
        self.lib = _lib.load()
        self.config = config
        self.codec_config = codec_config
        self.generate_defaults = generate_defaults or GenerateConfig()
        self.dtype = QTTS_BF16 if loader.dtype in ("bf16", "bfloat16") else QTTS_F32
        self._ctx = C.c_void_p()
        check(self.lib.qtts_init(loader.device, C.byref(self._ctx)))
        self._m = C.c_void_p()
        cj = json.dumps(codec_config).encode() if codec_config is not None else None
        check(self.lib.qtts_model_create(self._ctx, json.dumps(config).encode(), cj, self.dtype, C.byref(self._m)))
        self._finalized = False
        self.last_timing = None
        self.text_processor = None  # text.TextProcessor; set by ModelLoader when the directory holds tokenizer.json

    def open_session(self) -> "Model":
        """A second decode stream on this model (qtts_session_open): same weights, its own HIP stream, KV caches, workspaces
        and frame graph.  The returned object has every run-time method of Model (generate, decode_output, open_codec_stream,
        ...) and may be driven from another host thread; close it before closing the model."""
        if not self._finalized:
            raise QttsError(-5, "open_session: finalize the model first")
        ses = Model.__new__(Model)
        ses.lib = self.lib
        ses.config, ses.codec_config, ses.generate_defaults = self.config, self.codec_config, self.generate_defaults
        ses.dtype = self.dtype
        ses._ctx = C.c_void_p()          # borrowed from the parent: never shut down by the session
        ses._parent = self
        ses._m = C.c_void_p()
        check(self.lib.qtts_session_open(self._m, C.byref(ses._m)))
        ses._finalized = True
        ses.last_timing = None
        ses.text_processor = self.text_processor
        return ses

    def close(self):
        if getattr(self, "_m", None) is not None and self._m:
            if getattr(self, "_parent", None) is not None:
                self.lib.qtts_session_close(self._m)
            else:
                self.lib.qtts_model_free(self._m)
            self._m = C.c_void_p()
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.qtts_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ loading
    def load_tensor(self, key: str, arr):
        """Host numpy array (f32, or uint16 holding bf16 bits) or a CUDA torch tensor (f32 / bf16)."""
        if hasattr(arr, "is_cuda") and arr.is_cuda:
            import torch
            t = arr.contiguous()
            src = QTTS_BF16 if t.dtype == torch.bfloat16 else QTTS_F32
            if t.dtype not in (torch.bfloat16, torch.float32):
                raise QttsError(-1, "device tensors must be f32 or bf16")
            shape = (C.c_int64 * t.dim())(*t.shape)
            # the library copies on its own stream: the producer of `t` (torch's current stream) must be done first
            torch.cuda.current_stream(t.device).synchronize()
            check(self.lib.qtts_model_load_tensor_device(self._m, key.encode(), C.c_void_p(t.data_ptr()), src, shape, t.dim()))
            return
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint16:
            src = QTTS_BF16
        else:
            a = np.ascontiguousarray(a, np.float32)
            src = QTTS_F32
        shape = (C.c_int64 * a.ndim)(*a.shape)
        check(self.lib.qtts_model_load_tensor(self._m, key.encode(), a.ctypes.data_as(C.c_void_p), src, shape, a.ndim))

    def load_weights(self, items: Iterable):
        for k, a in items:
            self.load_tensor(k, a)

    def finalize(self):
        check(self.lib.qtts_model_finalize(self._m))
        self._finalized = True
        return self

    # ------------------------------------------------------------------ properties
    @property
    def hidden_size(self):
        return self.config["talker_config"]["hidden_size"]

    @property
    def num_code_groups(self):
        return self.config["talker_config"]["num_code_groups"]

    def sample_rate(self) -> int:  # model/generate.rs:196-201
        return (self.codec_config or {}).get("output_sample_rate", 24000)

    def total_upsample(self) -> int:
        d = self.codec_config["decoder_config"]
        return int(np.prod(d["upsample_rates"]) * np.prod(d["upsampling_ratios"]))

    # ------------------------------------------------------------------ validation (model/mod.rs:341-419)
    def model_type(self) -> str:
        return (self.config.get("tts_model_type") or "").lower()

    def validate_speaker(self, speaker: str):
        ids = self.config["talker_config"].get("spk_id") or {}
        if speaker.lower() not in ids:
            raise QttsError(-1, f"Unknown speaker '{speaker}'. Available: {sorted(ids)}")

    def validate_language(self, language: str):
        ids = self.config["talker_config"].get("codec_language_id") or {}
        if language.lower() != "auto" and language.lower() not in ids:
            raise QttsError(-1, f"Unsupported language '{language}'. Available: {['auto'] + sorted(ids)}")

    # ------------------------------------------------------------------ generation
    def generate(self, requests: Sequence[GenerationRequest], sampling: SamplingConfig, max_new_tokens: int,
                 trace: bool = False, use_graph: bool = True, profile_gemm: bool = False, first_frames: int = 0,
                 on_first_chunk=None, min_new_tokens: int = 0, slots: int = 0, shared_rng_stream: bool = False,
                 on_done=None) -> Output:
        """ConditionalGeneration::generate (nn/generation.rs:815) over a batch of independent utterances.
        slots > 0: continuous batching (qtts_generate_continuous) - the utterances go through at most `slots` rows, a finished
        row's slot is given to the next queued utterance, and every utterance gets the frames a batch of one would produce:
        Output.codes is padded to the longest one, Output.steps_per_utterance holds each one's own count.
        on_done(i, codes (num_steps, groups), effective_len), continuous batching only, is called as soon as utterance i has left
        its row; it may call decode_codes on this model (the stream is idle between frames), so finished utterances turn
        into audio while the other rows keep decoding.
        on_first_chunk(codes (batch, first_frames, groups)) is called once the first `first_frames` frames of every row
        exist, while generation continues afterwards (first-audio hook; the reference has no counterpart).
        min_new_tokens > 0 keeps EOS suppressed for that many frames (fixed-length throughput runs; 0 = the reference's 2).
        shared_rng_stream: all rows draw from one MT19937 stream (seed of the first request) in the order of the reference's
        global generator (nn/mt_rng.rs:22, nn/generation.rs:1055-1060) instead of one stream per request."""
        n = len(requests)
        G = self.num_code_groups
        V = self.config["talker_config"]["vocab_size"]
        H = self.hidden_size
        reqs = (_lib.Request * n)()
        keep = []
        for i, r in enumerate(requests):
            t = np.ascontiguousarray(r.text_ids, np.uint32)
            keep.append(t)
            reqs[i].text_ids = t.ctypes.data_as(C.POINTER(C.c_uint32))
            reqs[i].n_text = len(t)
            if r.instruct_ids is not None and len(r.instruct_ids):
                ins = np.ascontiguousarray(r.instruct_ids, np.uint32)
                keep.append(ins)
                reqs[i].instruct_ids = ins.ctypes.data_as(C.POINTER(C.c_uint32))
                reqs[i].n_instruct = len(ins)
            reqs[i].language = (r.language or "auto").encode()
            reqs[i].speaker = r.speaker.encode() if r.speaker else None
            if r.speaker_embed is not None:
                se = np.ascontiguousarray(r.speaker_embed, np.float32).reshape(-1)
                if se.size != H:
                    raise QttsError(-4, "speaker_embed must have hidden_size elements")
                keep.append(se)
                reqs[i].speaker_embed = se.ctypes.data_as(C.POINTER(C.c_float))
            if r.ref_codes is not None:
                rc = np.ascontiguousarray(r.ref_codes, np.uint32)
                keep.append(rc)
                reqs[i].ref_codes = rc.ctypes.data_as(C.POINTER(C.c_uint32))
                reqs[i].ref_T = rc.shape[0]
            if r.ref_text_ids is not None:
                rt = np.ascontiguousarray(r.ref_text_ids, np.uint32)
                keep.append(rt)
                reqs[i].ref_text_ids = rt.ctypes.data_as(C.POINTER(C.c_uint32))
                reqs[i].n_ref_text = len(rt)
            reqs[i].non_streaming = int(r.non_streaming_mode)
            reqs[i].use_seed = 0 if r.seed is None else 1
            reqs[i].seed = 0 if r.seed is None else int(r.seed)
            reqs[i].max_new_tokens = int(r.max_new_tokens or 0)
        codes = np.zeros((n, max_new_tokens, G), np.uint32)
        outs = (_lib.CodesOut * n)()
        for i in range(n):
            outs[i].codes = codes[i].ctypes.data_as(C.POINTER(C.c_uint32))
        sc = sampling.to_c()
        dbg = _lib.GenDebug()
        logits = hidden = cp_logits = None
        if trace:
            Vc = self.config["talker_config"]["code_predictor_config"]["vocab_size"]
            logits = np.zeros((n, max_new_tokens, V), np.float32)
            hidden = np.zeros((n, max_new_tokens, H), np.float32)
            cp_logits = np.zeros((n, max_new_tokens, G - 1, Vc), np.float32)
            dbg.logits = logits.ctypes.data_as(C.POINTER(C.c_float))
            dbg.hidden = hidden.ctypes.data_as(C.POINTER(C.c_float))
            dbg.cp_logits = cp_logits.ctypes.data_as(C.POINTER(C.c_float))
        dbg.use_graph = int(use_graph)
        dbg.profile_gemm = int(profile_gemm)
        dbg.min_new_tokens = int(min_new_tokens)
        dbg.shared_rng_stream = int(bool(shared_rng_stream))
        cb_err = []
        if on_first_chunk is not None and first_frames > 0:
            def _cb(_user, ptr, nrows, frames):
                try:  # exceptions must not unwind through the C frames
                    on_first_chunk(np.ctypeslib.as_array(ptr, shape=(nrows, frames, G)).copy())
                except Exception as ex:
                    cb_err.append(ex)
            cb = _lib.CHUNK_CB(_cb)
            keep.append(cb)
            dbg.first_frames = int(first_frames)
            dbg.chunk_cb = cb
        if on_done is not None:
            if slots <= 0:
                raise QttsError(-1, "on_done needs continuous batching (slots > 0)")

            def _done(_user, u, ptr, num_steps, eff):
                try:  # exceptions must not unwind through the C frames
                    c = np.ctypeslib.as_array(ptr, shape=(num_steps, G)).copy() if num_steps > 0 else np.zeros((0, G), np.uint32)
                    on_done(int(u), c, int(eff))
                except Exception as ex:
                    cb_err.append(ex)
            dcb = _lib.DONE_CB(_done)
            keep.append(dcb)
            dbg.done_cb = dcb
        if slots > 0:
            if trace or on_first_chunk is not None:
                raise QttsError(-1, "continuous batching has no traces and no first-chunk hook")
            check(self.lib.qtts_generate_continuous(self._m, reqs, n, C.byref(sc), max_new_tokens, int(slots), outs, C.byref(dbg)))
        else:
            check(self.lib.qtts_generate_ex(self._m, reqs, n, C.byref(sc), max_new_tokens, outs, C.byref(dbg)))
        if cb_err:
            raise cb_err[0]
        tm = (C.c_double * 8)()
        check(self.lib.qtts_last_timing(self._m, tm, 8))
        self.last_timing = dict(prefill_ms=tm[0], frames_ms=tm[1], gemm_ms=tm[2], gemm_launches=int(tm[3]),
                                gemm_bytes=tm[4], frames=int(tm[5]), graph_launch_host_ms=tm[7])
        steps = max(outs[i].num_steps for i in range(n)) if slots > 0 else outs[0].num_steps
        out = Output(codes=codes[:, :steps].copy(), num_steps=steps, effective_lengths=[outs[i].effective_len for i in range(n)])
        out.steps_per_utterance = [outs[i].num_steps for i in range(n)]
        if trace:
            out.logits = logits[:, :steps]
            out.hidden = hidden[:, :steps]
            out.cp_logits = cp_logits[:, :steps]
        return out

    def gemm_profile(self):
        """Per-shape GEMM totals of the last generate(..., use_graph=False, profile_gemm=True):
        list of dicts (M, N, K, launches, total_us)."""
        rows = (C.c_double * (5 * 64))()
        n = C.c_int(0)
        check(self.lib.qtts_gemm_profile(self._m, rows, 64, C.byref(n)))
        return [dict(M=int(rows[5 * i]), N=int(rows[5 * i + 1]), K=int(rows[5 * i + 2]), launches=int(rows[5 * i + 3]),
                     total_us=rows[5 * i + 4]) for i in range(min(n.value, 64))]

    def bench_gemm(self, M: int, N: int, K: int, iters: int = 50) -> float:
        """Average microseconds of the decode path's weight-streaming GEMM for one shape (back-to-back launches)."""
        us = C.c_double(0)
        check(self.lib.qtts_bench_gemm(self._ctx, self.dtype, M, N, K, 0, iters, C.byref(us)))
        return us.value

    def codec_device_ms(self) -> float:
        tm = (C.c_double * 8)()
        check(self.lib.qtts_last_timing(self._m, tm, 8))
        return tm[6]

    def decode_codes(self, codes: np.ndarray, lens: Optional[Sequence[int]] = None, chunk: int = 0, left_ctx: int = 0):
        """TokenizerV2::decode / chunked_decode on (batch, steps, quantizers) codes; returns a list of waveforms."""
        c = np.ascontiguousarray(codes, np.uint32)
        if c.ndim == 2:
            c = c[None]
        B, T, Q = c.shape
        lens_a = np.ascontiguousarray(lens if lens is not None else [T] * B, np.int32)
        stride = (T + 1) * self.total_upsample()
        wav = np.empty((B, stride), np.float32)   # rows are valid up to wl[b]; nothing past that is handed out
        wl = np.zeros(B, np.int64)
        check(self.lib.qtts_codec_decode(self._m, c.ctypes.data_as(C.c_void_p), lens_a.ctypes.data_as(C.c_void_p), B, T, chunk,
                                         left_ctx, wav.ctypes.data_as(C.c_void_p), stride, wl.ctypes.data_as(C.c_void_p)))
        # views of the one host buffer the library filled (a second pass over a batch's waveforms costs tens of milliseconds)
        return [wav[b, :wl[b]] for b in range(B)]

    def decode_output(self, output: Output):
        """Model::decode_output (model/generate.rs:29-143): truncate to effective lengths, chunk long sequences."""
        return self.decode_codes(output.codes, output.effective_lengths, chunk=-2, left_ctx=0)

    def open_codec_stream(self, batch: int, max_chunk_frames: int) -> "CodecStream":
        """Streaming decode session: the concatenated pushes equal one whole `decode_codes` of the concatenated codes."""
        return CodecStream(self, batch, max_chunk_frames)

    def encode_audio(self, audio: Sequence[np.ndarray], num_quantizers: Optional[int] = None):
        """Model::encode_audio / TokenizerV2::encode (audio/tokenizer/v2/mod.rs:164-171): 24 kHz clips -> list of
        (frames, num_quantizers) code arrays, frames = samples // 1920."""
        if self.codec_config is None:
            raise QttsError(-5, "Audio tokenizer not loaded")
        Q = int(num_quantizers or self.codec_config.get("encoder_valid_num_quantizers", 16))
        B = len(audio)
        ns = np.array([len(a) for a in audio], np.int32)
        mx = int(ns.max())
        buf = np.zeros((B, mx), np.float32)
        for b, a in enumerate(audio):
            buf[b, :len(a)] = a
        cap = mx // 1920 + 1
        codes = np.zeros((B, cap, Q), np.uint32)
        tl = np.zeros(B, np.int32)
        check(self.lib.qtts_encode_audio(self._m, buf.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), B, mx, Q,
                                         codes.ctypes.data_as(C.c_void_p), cap, tl.ctypes.data_as(C.c_void_p)))
        return [codes[b, :tl[b]].copy() for b in range(B)]

    def create_voice_clone_prompt_from_audio(self, audio: np.ndarray, ref_text_ids=None, x_vector_only: bool = False):
        """Model::create_voice_clone_prompt_from_audio (model/mod.rs:643-700): speaker embedding from the clip and, in ICL
        mode, its codes; the transcript arrives tokenised (tokenize_ref_text output)."""
        emb = self.extract_speaker_embedding([audio])[0]
        if x_vector_only:
            return VoiceClonePromptItem.x_vector_only(emb)
        if ref_text_ids is None:
            raise QttsError(-5, "ICL mode requires ref_text to be present")
        return VoiceClonePromptItem.icl(self.encode_audio([audio])[0], emb, ref_text_ids)

    def compute_mel_spectrogram(self, audio: Sequence[np.ndarray], no_alloc: bool = False):
        """Model::compute_mel_spectrogram (model/mod.rs:520-528): list of (frames, 128) arrays.
        no_alloc: go through qtts_mel_noalloc (fixed device staging array, no device allocation, bounded batch size)."""
        B = len(audio)
        ns = np.array([len(a) for a in audio], np.int32)
        mx = int(ns.max())
        buf = np.zeros((B, mx), np.float32)
        for b, a in enumerate(audio):
            buf[b, :len(a)] = a
        fs = mx // 256 + 8
        mel = np.zeros((B, fs, 128), np.float32)
        nf = np.zeros(B, np.int32)
        fn = self.lib.qtts_mel_noalloc if no_alloc else self.lib.qtts_mel
        check(fn(self._m, buf.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), B, mx,
                 mel.ctypes.data_as(C.c_void_p), fs, nf.ctypes.data_as(C.c_void_p)))
        return [mel[b, :nf[b]].copy() for b in range(B)]

    def extract_speaker_embedding(self, audio: Sequence[np.ndarray]) -> np.ndarray:
        """mel + ConditionalGeneration::encode_speaker (model/mod.rs:560-600): (batch, enc_dim)."""
        s = self.config.get("speaker_encoder_config")
        if not s:
            raise QttsError(-5, "Speaker encoder not available - this model doesn't support voice cloning. "
                                "Use a Base model for voice cloning.")
        B = len(audio)
        ns = np.array([len(a) for a in audio], np.int32)
        mx = int(ns.max())
        buf = np.zeros((B, mx), np.float32)
        for b, a in enumerate(audio):
            buf[b, :len(a)] = a
        emb = np.zeros((B, s["enc_dim"]), np.float32)
        check(self.lib.qtts_speaker_embed(self._m, buf.ctypes.data_as(C.c_void_p), ns.ctypes.data_as(C.c_void_p), B, mx,
                                          emb.ctypes.data_as(C.c_void_p)))
        return emb

    # ---- high-level API (model/generate.rs:227-848); text arrives tokenised (the tokenizer stays on the host side)
    def _finish(self, out: Output, decode: bool) -> GenerationResult:
        if out.num_steps == 0:
            raise QttsError(-5, "Generation produced no audio. The model generated 0 steps.")
        audio = self.decode_output(out) if decode and self.codec_config is not None else []
        return GenerationResult(audio=audio, sample_rate=self.sample_rate(), codes=out.codes,
                                effective_lengths=out.effective_lengths)

    def generate_custom_voice_full(self, text_ids, speaker: str, language: str, instruct_ids=None, options: dict = None,
                                   decode: bool = True) -> GenerationResult:
        if self.model_type() not in ("custom_voice", "customvoice", ""):
            raise QttsError(-5, f"Model type '{self.model_type()}' does not support custom voice generation")
        self.validate_speaker(speaker)
        self.validate_language(language)
        o = dict(options or {})
        sampling = self.generate_defaults.merge(**o)
        mx = self.generate_defaults.effective_max_tokens(o.get("max_new_tokens"))
        req = GenerationRequest(text_ids=text_ids, language=language, speaker=speaker, instruct_ids=instruct_ids,
                                non_streaming_mode=o.get("non_streaming_mode", True) is not False, seed=o.get("seed"))
        return self._finish(self.generate([req], sampling, mx), decode)

    def generate_voice_design_full(self, text_ids, instruct_ids, language: str, options: dict = None,
                                   decode: bool = True) -> GenerationResult:
        self.validate_language(language)
        o = dict(options or {})
        sampling = self.generate_defaults.merge(**o)
        mx = self.generate_defaults.effective_max_tokens(o.get("max_new_tokens"))
        req = GenerationRequest(text_ids=text_ids, language=language, instruct_ids=instruct_ids,
                                non_streaming_mode=o.get("non_streaming_mode", True) is not False, seed=o.get("seed"))
        return self._finish(self.generate([req], sampling, mx), decode)

    def generate_voice_clone_full(self, text_ids, language: str, prompt, options: dict = None,
                                  decode: bool = True) -> GenerationResult:
        """Voice clone (model/generate.rs:452-644).  `prompt` is a VoiceClonePromptItem (or a bare speaker embedding for
        x-vector-only mode).  ICL mode feeds the reference codes / transcript into the prompt, decodes
        [ref_code || generated codes] and cuts floor(ref_len / total_len * n_samples) leading samples."""
        self.validate_language(language)
        if not isinstance(prompt, VoiceClonePromptItem):
            prompt = VoiceClonePromptItem.x_vector_only(np.asarray(prompt, np.float32))
        prompt.validate()
        o = dict(options or {})
        sampling = self.generate_defaults.merge(**o)
        mx = self.generate_defaults.effective_max_tokens(o.get("max_new_tokens"))
        ns = o.get("non_streaming_mode")
        req = GenerationRequest(text_ids=text_ids, language=language, speaker_embed=prompt.ref_spk_embedding,
                                ref_codes=prompt.ref_code, ref_text_ids=prompt.ref_text_ids,
                                non_streaming_mode=(not prompt.x_vector_only_mode) if ns is None else bool(ns), seed=o.get("seed"))
        out = self.generate([req], sampling, mx)
        if out.num_steps == 0:
            raise QttsError(-5, "Generation produced no audio. The model generated 0 steps.")
        audio = []
        if decode and self.codec_config is not None:
            if prompt.ref_code is not None and not prompt.x_vector_only_mode:
                ref = np.ascontiguousarray(prompt.ref_code, np.uint32)
                comb = np.concatenate([ref[None], out.codes], axis=1)
                total = out.effective_lengths[0] + len(ref)
                full = self.decode_output(Output(codes=comb, num_steps=out.num_steps + len(ref), effective_lengths=[total]))[0]
                cut = int(len(ref) / total * len(full))
                audio = [full[cut:]]
            else:
                audio = self.decode_output(out)
        return GenerationResult(audio=audio, sample_rate=self.sample_rate(), codes=out.codes, effective_lengths=out.effective_lengths)

    # ---- string front ends (model/mod.rs:341-365, model/generate.rs:680-794): tokenise, then the *_full methods
    def tokenize_text(self, text: str):
        if self.text_processor is None:
            raise QttsError(-5, "No text processor loaded. Load a tokenizer.json file first.")
        return self.text_processor.tokenize_for_tts(text)

    def tokenize_texts(self, texts, padding_side: str = "left"):
        if self.text_processor is None:
            raise QttsError(-5, "No text processor loaded. Load a tokenizer.json file first.")
        return self.text_processor.batch_tokenize_for_tts(texts, padding_side)

    def generate_custom_voice_from_text(self, text: str, speaker: str, language: str, instruct: Optional[str] = None,
                                        options: dict = None, decode: bool = True) -> GenerationResult:
        ins = self.text_processor.tokenize_instruct(instruct) if instruct and self.text_processor else None
        return self.generate_custom_voice_full(self.tokenize_text(text), speaker, language, ins, options, decode)

    def generate_voice_design_from_text(self, text: str, voice_description: str, language: str, options: dict = None,
                                        decode: bool = True) -> GenerationResult:
        ids = self.tokenize_text(text)
        return self.generate_voice_design_full(ids, self.text_processor.tokenize_instruct(voice_description), language, options, decode)

    def generate_voice_clone_from_text(self, text: str, prompt, language: str, options: dict = None, decode: bool = True) -> GenerationResult:
        return self.generate_voice_clone_full(self.tokenize_text(text), language, prompt, options, decode)

    def encode_speaker_from_mel(self, mels: Sequence[np.ndarray]) -> np.ndarray:
        """ConditionalGeneration::encode_speaker (nn/generation.rs:140) on (frames, 128) log-mels: (batch, enc_dim)."""
        s = self.config.get("speaker_encoder_config")
        if not s:
            raise QttsError(-5, "Speaker encoder not available - this model doesn't support voice cloning. "
                                "Use a Base model for voice cloning.")
        B = len(mels)
        nf = np.array([len(x) for x in mels], np.int32)
        stride = int(nf.max())
        buf = np.zeros((B, stride, 128), np.float32)
        for b, x in enumerate(mels):
            buf[b, :len(x)] = x
        emb = np.zeros((B, s["enc_dim"]), np.float32)
        check(self.lib.qtts_speaker_embed_mel(self._m, buf.ctypes.data_as(C.c_void_p), nf.ctypes.data_as(C.c_void_p), B, stride,
                                              emb.ctypes.data_as(C.c_void_p)))
        return emb

    def create_voice_clone_prompt_from_mel(self, mel: np.ndarray, ref_text_ids=None, x_vector_only: bool = True):
        """Model::create_voice_clone_prompt_from_mel (model/mod.rs:579-612): speaker embedding from a (frames, 128) mel.  Like
        the reference, the ICL variant built from a mel alone carries no reference codes."""
        emb = self.encode_speaker_from_mel([np.asarray(mel, np.float32).reshape(-1, 128)])[0]
        if x_vector_only:
            return VoiceClonePromptItem.x_vector_only(emb)
        return VoiceClonePromptItem(ref_code=None, ref_spk_embedding=emb, x_vector_only_mode=False, ref_text_ids=ref_text_ids)

    def generate_custom_voice_from_texts_batch(self, text_ids_list, speaker: str, language: str, options: dict = None,
                                               decode: bool = True) -> GenerationResult:
        """model/generate.rs:809-848 loops utterance by utterance; here the rows share one batched decode."""
        self.validate_speaker(speaker)
        self.validate_language(language)
        o = dict(options or {})
        sampling = self.generate_defaults.merge(**o)
        mx = self.generate_defaults.effective_max_tokens(o.get("max_new_tokens"))
        seed = o.get("seed")
        reqs = [GenerationRequest(text_ids=t, language=language, speaker=speaker,
                                  seed=None if seed is None else seed + i) for i, t in enumerate(text_ids_list)]
        return self._finish(self.generate(reqs, sampling, mx), decode)


class CodecStream:
    """Handle of a streaming decode session (qtts_codec_stream_*).  The reference has no counterpart: its chunked decode
    restarts every chunk with 32 frames of context and is not equal to a whole decode."""

    def __init__(self, model: "Model", batch: int, max_chunk_frames: int):
        self.model, self.batch, self.max_chunk = model, int(batch), int(max_chunk_frames)
        self._s = C.c_void_p()
        check(model.lib.qtts_codec_stream_open(model._m, self.batch, self.max_chunk, C.byref(self._s)))

    def push(self, codes: np.ndarray) -> np.ndarray:
        """codes (batch, n, quantizers): the next n frames of every row -> (batch, samples) decided by them."""
        if self._s is None:
            raise QttsError(-5, "codec stream is closed")
        c = np.ascontiguousarray(codes, np.uint32)
        if c.ndim == 2:
            c = c[None]
        B, n, Q = c.shape
        if B != self.batch:
            raise QttsError(-1, f"codec stream was opened for batch {self.batch}, got {B}")
        stride = n * self.model.total_upsample()
        wav = np.empty((B, stride), np.float32)
        ns = C.c_int64(0)
        check(self.model.lib.qtts_codec_stream_push(self._s, c.ctypes.data_as(C.c_void_p), n, wav.ctypes.data_as(C.c_void_p), stride,
                                                     C.byref(ns)))
        return wav[:, :ns.value]

    def close(self):
        if self._s is not None:
            self.model.lib.qtts_codec_stream_close(self._s)
            self._s = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class ModelLoader:
    """ModelLoader (model/loader.rs:109-118): reads config.json, speech_tokenizer/config.json,
    generate_config.json and the safetensors files of a model directory."""

    def __init__(self, model_dir: str):
        self.model_dir = model_dir
        with open(os.path.join(model_dir, "config.json")) as f:
            self.config = json.load(f)
        cpath = os.path.join(model_dir, "speech_tokenizer", "config.json")
        self.codec_config = json.load(open(cpath)) if os.path.exists(cpath) else None
        gpath = os.path.join(model_dir, "generate_config.json")
        self.generate_defaults = GenerateConfig.from_json(open(gpath).read()) if os.path.exists(gpath) else GenerateConfig()

    @classmethod
    def from_local_dir(cls, model_dir: str) -> "ModelLoader":
        if not os.path.exists(os.path.join(model_dir, "config.json")):
            raise FileNotFoundError(f"config.json not found in {model_dir}")
        return cls(model_dir)

    def _tensor_files(self):
        # like the reference, only the first shard of a sharded checkpoint is opened (model/loader.rs:219-235)
        main = os.path.join(self.model_dir, "model.safetensors")
        if not os.path.exists(main):
            shards = sorted(f for f in os.listdir(self.model_dir) if f.startswith("model-") and f.endswith(".safetensors"))
            if not shards:
                raise FileNotFoundError("no model.safetensors in " + self.model_dir)
            main = os.path.join(self.model_dir, shards[0])
        files = [main]
        tok = os.path.join(self.model_dir, "speech_tokenizer", "model.safetensors")
        if os.path.exists(tok):
            files.append(tok)
        return files

    def load_tts_model(self, loader: LoaderConfig = LoaderConfig()) -> Model:
        from safetensors import safe_open
        m = Model(self.config, self.codec_config if loader.load_tokenizer else None, loader, self.generate_defaults)
        if loader.load_text_tokenizer and os.path.exists(os.path.join(self.model_dir, "tokenizer.json")):
            from .text import TextProcessor
            m.text_processor = TextProcessor.from_pretrained(self.model_dir)
        for path in self._tensor_files():
            with safe_open(path, framework="np") as f:
                for key in f.keys():
                    if key.startswith("encoder.") and not loader.load_encoder:
                        continue
                    try:
                        arr = f.get_tensor(key)
                    except TypeError:  # bf16 is not a numpy dtype: read the raw bits through torch
                        import torch
                        from safetensors import safe_open as so
                        with so(path, framework="pt") as ft:
                            arr = ft.get_tensor(key).view(torch.uint16).numpy()
                    m.load_tensor(key, arr)
        return m.finalize()


    def load_tts_model_native(self, loader: LoaderConfig = LoaderConfig()) -> Model:
        """The same model through the library's own file reader (qtts_model_load_dir): configs and safetensors files are read
        and mapped by the C++ side, nothing passes through numpy.  What a Rust caller of the C ABI would use."""
        m = Model.__new__(Model)
        m.lib = _lib.load()
        m.config, m.codec_config = self.config, (self.codec_config if loader.load_tokenizer else None)
        m.generate_defaults = self.generate_defaults
        m.dtype = QTTS_BF16 if loader.dtype in ("bf16", "bfloat16") else QTTS_F32
        m._ctx = C.c_void_p()
        m._m = C.c_void_p()
        m._finalized = False
        m.last_timing = None
        m.text_processor = None
        check(m.lib.qtts_init(loader.device, C.byref(m._ctx)))
        check(m.lib.qtts_model_load_dir(m._ctx, os.fsencode(self.model_dir), m.dtype, 1 if loader.load_tokenizer else 0,
                                        1 if loader.load_encoder else 0, C.byref(m._m)))
        m._finalized = True
        if loader.load_text_tokenizer and os.path.exists(os.path.join(self.model_dir, "tokenizer.json")):
            from .text import TextProcessor
            m.text_processor = TextProcessor.from_pretrained(self.model_dir)
        return m


def synthetic_model(shape: str, dtype: str = "bf16", seed: int = 20261111, model_type: str = "custom_voice", codec: bool = True,
                    talker: bool = True, device: int = 0, round_to_bf16: bool = False, encoder: bool = False):
    """Random-init model of a named shape set (see synth.py).  Returns (Model, config, codec_config, template)."""
    from . import synth
    cfg, ccfg, tmpl = synth.shape_set(shape, model_type)
    m = Model(cfg, ccfg if codec else None, LoaderConfig(dtype=dtype, device=device))
    keys = []
    if talker:
        keys += synth.talker_keys(cfg) + synth.speaker_keys(cfg)
    if codec:
        keys += synth.codec_keys(ccfg)
    if encoder:
        keys += synth.encoder_keys()
    for k, a in synth.iter_weights(keys, seed):
        if round_to_bf16:
            a = round_bf16(a)
        m.load_tensor(k, a)
    m.finalize()
    return m, cfg, ccfg, tmpl


def round_bf16(a: np.ndarray) -> np.ndarray:
    """f32 -> nearest-even bf16 -> f32 (so that a bf16 run and its f32 reference share identical weights)."""
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000))
    return r.view(np.float32).reshape(a.shape)
