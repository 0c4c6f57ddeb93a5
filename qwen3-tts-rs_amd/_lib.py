"""ctypes binding of libqtts_hip.so (the C ABI in include/qtts.h).  Fails loudly when the HIP library is
missing: there is no CPU fallback on the product path."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqtts_hip.so")

QTTS_F32, QTTS_BF16 = 0, 1


class QttsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"qtts error {code}: {msg}")
        self.code = code


class Request(C.Structure):
    _fields_ = [("text_ids", C.POINTER(C.c_uint32)), ("n_text", C.c_int),
                ("instruct_ids", C.POINTER(C.c_uint32)), ("n_instruct", C.c_int),
                ("language", C.c_char_p), ("speaker", C.c_char_p),
                ("speaker_embed", C.POINTER(C.c_float)),
                ("ref_codes", C.POINTER(C.c_uint32)), ("ref_T", C.c_int),
                ("ref_text_ids", C.POINTER(C.c_uint32)), ("n_ref_text", C.c_int),
                ("non_streaming", C.c_int), ("use_seed", C.c_int), ("seed", C.c_uint64),
                ("max_new_tokens", C.c_int)]


class Sampling(C.Structure):
    _fields_ = [("temperature", C.c_double), ("top_k", C.c_int), ("top_p", C.c_double), ("do_sample", C.c_int),
                ("repetition_penalty", C.c_double), ("subtalker_do_sample", C.c_int),
                ("subtalker_temperature", C.c_double), ("subtalker_top_k", C.c_int), ("subtalker_top_p", C.c_double),
                ("min_new_tokens", C.c_int)]


class CodesOut(C.Structure):
    _fields_ = [("codes", C.POINTER(C.c_uint32)), ("num_steps", C.c_int), ("effective_len", C.c_int)]


CHUNK_CB = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_int)
DONE_CB = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int)


class GenDebug(C.Structure):
    _fields_ = [("logits", C.POINTER(C.c_float)), ("hidden", C.POINTER(C.c_float)), ("use_graph", C.c_int),
                ("profile_gemm", C.c_int), ("first_frames", C.c_int), ("chunk_cb", CHUNK_CB), ("chunk_user", C.c_void_p),
                ("min_new_tokens", C.c_int), ("cp_logits", C.POINTER(C.c_float)), ("shared_rng_stream", C.c_int),
                ("done_cb", DONE_CB), ("done_user", C.c_void_p)]


EXPORTS = ["qtts_init", "qtts_shutdown", "qtts_last_error", "qtts_stream", "qtts_model_create", "qtts_model_load_tensor",
           "qtts_model_load_tensor_device", "qtts_model_finalize", "qtts_model_load_dir", "qtts_model_free", "qtts_generate", "qtts_generate_ex",
           "qtts_last_timing", "qtts_codec_decode", "qtts_mel", "qtts_speaker_embed", "qtts_model_info",
           "qtts_gemm_profile", "qtts_bench_gemm", "qtts_bench_stream", "qtts_encode_audio",
           "qtts_speaker_embed_mel", "qtts_codec_stream_open", "qtts_codec_stream_push", "qtts_codec_stream_close",
           "qtts_mel_noalloc", "qtts_mel_noalloc_capacity", "qtts_session_open", "qtts_session_close", "qtts_session_stream",
           "qtts_generate_continuous"]

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    lib.qtts_last_error.restype = C.c_char_p
    lib.qtts_stream.restype = C.c_void_p
    lib.qtts_stream.argtypes = [C.c_void_p]
    lib.qtts_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.qtts_shutdown.argtypes = [C.c_void_p]
    lib.qtts_shutdown.restype = None
    lib.qtts_model_create.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]
    lib.qtts_model_load_tensor.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_int]
    lib.qtts_model_load_tensor_device.argtypes = lib.qtts_model_load_tensor.argtypes
    lib.qtts_model_finalize.argtypes = [C.c_void_p]
    lib.qtts_model_load_dir.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.qtts_model_free.argtypes = [C.c_void_p]
    lib.qtts_model_free.restype = None
    lib.qtts_session_open.argtypes = [C.c_void_p, C.POINTER(C.c_void_p)]
    lib.qtts_session_close.argtypes = [C.c_void_p]
    lib.qtts_session_close.restype = None
    lib.qtts_session_stream.argtypes = [C.c_void_p]
    lib.qtts_session_stream.restype = C.c_void_p
    lib.qtts_generate.argtypes = [C.c_void_p, C.POINTER(Request), C.c_int, C.POINTER(Sampling), C.c_int, C.POINTER(CodesOut)]
    lib.qtts_generate_ex.argtypes = lib.qtts_generate.argtypes + [C.POINTER(GenDebug)]
    lib.qtts_generate_continuous.argtypes = [C.c_void_p, C.POINTER(Request), C.c_int, C.POINTER(Sampling), C.c_int, C.c_int,
                                             C.POINTER(CodesOut), C.POINTER(GenDebug)]
    lib.qtts_last_timing.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int]
    lib.qtts_codec_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int64, C.c_void_p]
    lib.qtts_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.qtts_mel_noalloc.argtypes = lib.qtts_mel.argtypes
    lib.qtts_mel_noalloc_capacity.restype = C.c_int64
    lib.qtts_mel_noalloc_capacity.argtypes = []
    lib.qtts_speaker_embed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.qtts_speaker_embed_mel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    lib.qtts_encode_audio.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    lib.qtts_codec_stream_open.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.qtts_codec_stream_push.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]
    lib.qtts_codec_stream_close.argtypes = [C.c_void_p]
    lib.qtts_codec_stream_close.restype = None
    lib.qtts_model_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 5
    lib.qtts_gemm_profile.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int)]
    lib.qtts_bench_gemm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lib.qtts_bench_stream.argtypes = [C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise QttsError(rc, load().qtts_last_error().decode(errors="replace"))
