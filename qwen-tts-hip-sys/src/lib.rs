//! Raw bindings to `libqtts_hip.so` — one declaration per entry point of `include/qtts.h`, in the header's order, with the
//! header's struct layouts (`#[repr(C)]`).  Every function returns 0 or a negative `QTTS_ERR_*` class; the message of the
//! last failure on the calling thread is `qtts_last_error()`.  Nothing unwinds across the boundary.
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_double, c_float, c_int, c_long, c_void};

pub const QTTS_ERR_BAD_ARG: c_int = -1;
pub const QTTS_ERR_HIP: c_int = -2;
pub const QTTS_ERR_OOM: c_int = -3;
pub const QTTS_ERR_SHAPE: c_int = -4;
pub const QTTS_ERR_STATE: c_int = -5;
pub const QTTS_ERR_IO: c_int = -6;

pub const QTTS_F32: c_int = 0;
pub const QTTS_BF16: c_int = 1;

#[repr(C)]
pub struct qtts_ctx {
    _p: [u8; 0],
}
#[repr(C)]
pub struct qtts_model {
    _p: [u8; 0],
}
/// A decode session is a `qtts_model` handle that borrows another model's weights (`qtts_session_open`).
pub type qtts_session = qtts_model;
#[repr(C)]
pub struct qtts_codec_stream {
    _p: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct qtts_request {
    pub text_ids: *const u32,
    pub n_text: c_int,
    pub instruct_ids: *const u32,
    pub n_instruct: c_int,
    pub language: *const c_char,
    pub speaker: *const c_char,
    pub speaker_embed: *const c_float,
    pub ref_codes: *const u32,
    pub ref_T: c_int,
    pub ref_text_ids: *const u32,
    pub n_ref_text: c_int,
    pub non_streaming: c_int,
    pub use_seed: c_int,
    pub seed: u64,
    pub max_new_tokens: c_int,
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct qtts_sampling {
    pub temperature: c_double,
    pub top_k: c_int,
    pub top_p: c_double,
    pub do_sample: c_int,
    pub repetition_penalty: c_double,
    pub subtalker_do_sample: c_int,
    pub subtalker_temperature: c_double,
    pub subtalker_top_k: c_int,
    pub subtalker_top_p: c_double,
    pub min_new_tokens: c_int,
}

#[repr(C)]
pub struct qtts_codes_out {
    pub codes: *mut u32,
    pub num_steps: c_int,
    pub effective_len: c_int,
}

pub type qtts_chunk_cb = Option<unsafe extern "C" fn(user: *mut c_void, codes: *const u32, n: c_int, frames: c_int)>;
pub type qtts_done_cb = Option<
    unsafe extern "C" fn(user: *mut c_void, utterance: c_int, codes: *const u32, num_steps: c_int, effective_len: c_int),
>;

#[repr(C)]
pub struct qtts_gen_debug {
    pub logits: *mut c_float,
    pub hidden: *mut c_float,
    pub use_graph: c_int,
    pub profile_gemm: c_int,
    pub first_frames: c_int,
    pub chunk_cb: qtts_chunk_cb,
    pub chunk_user: *mut c_void,
    pub min_new_tokens: c_int,
    pub cp_logits: *mut c_float,
    pub shared_rng_stream: c_int,
    pub done_cb: qtts_done_cb,
    pub done_user: *mut c_void,
}

extern "C" {
    pub fn qtts_init(device_ordinal: c_int, out: *mut *mut qtts_ctx) -> c_int;
    pub fn qtts_shutdown(ctx: *mut qtts_ctx);
    pub fn qtts_last_error() -> *const c_char;
    pub fn qtts_stream(ctx: *mut qtts_ctx) -> *mut c_void;

    pub fn qtts_model_create(
        ctx: *mut qtts_ctx,
        config_json: *const c_char,
        codec_config_json: *const c_char,
        dtype: c_int,
        out: *mut *mut qtts_model,
    ) -> c_int;
    pub fn qtts_model_load_tensor(
        m: *mut qtts_model,
        hf_key: *const c_char,
        data: *const c_void,
        src_dtype: c_int,
        shape: *const i64,
        rank: c_int,
    ) -> c_int;
    pub fn qtts_model_load_tensor_device(
        m: *mut qtts_model,
        hf_key: *const c_char,
        data: *const c_void,
        src_dtype: c_int,
        shape: *const i64,
        rank: c_int,
    ) -> c_int;
    pub fn qtts_model_finalize(m: *mut qtts_model) -> c_int;
    pub fn qtts_model_load_dir(
        ctx: *mut qtts_ctx,
        model_dir: *const c_char,
        dtype: c_int,
        load_tokenizer: c_int,
        load_encoder: c_int,
        out: *mut *mut qtts_model,
    ) -> c_int;
    pub fn qtts_model_free(m: *mut qtts_model);

    pub fn qtts_session_open(m: *mut qtts_model, out: *mut *mut qtts_session) -> c_int;
    pub fn qtts_session_close(s: *mut qtts_session);
    pub fn qtts_session_stream(s: *mut qtts_session) -> *mut c_void;

    pub fn qtts_generate(
        m: *mut qtts_model,
        reqs: *const qtts_request,
        n: c_int,
        s: *const qtts_sampling,
        max_new_tokens: c_int,
        outs: *mut qtts_codes_out,
    ) -> c_int;
    pub fn qtts_generate_ex(
        m: *mut qtts_model,
        reqs: *const qtts_request,
        n: c_int,
        s: *const qtts_sampling,
        max_new_tokens: c_int,
        outs: *mut qtts_codes_out,
        dbg: *const qtts_gen_debug,
    ) -> c_int;
    pub fn qtts_generate_continuous(
        m: *mut qtts_model,
        reqs: *const qtts_request,
        n: c_int,
        s: *const qtts_sampling,
        max_new_tokens: c_int,
        slots: c_int,
        outs: *mut qtts_codes_out,
        dbg: *const qtts_gen_debug,
    ) -> c_int;
    pub fn qtts_last_timing(m: *mut qtts_model, out: *mut c_double, n: c_int) -> c_int;

    pub fn qtts_gemm_profile(m: *mut qtts_model, rows: *mut c_double, cap_rows: c_int, n_rows: *mut c_int) -> c_int;
    pub fn qtts_bench_gemm(
        ctx: *mut qtts_ctx,
        dtype: c_int,
        M: c_int,
        N: c_int,
        K: c_int,
        KS: c_int,
        iters: c_int,
        avg_us: *mut c_double,
    ) -> c_int;
    pub fn qtts_bench_stream(
        ctx: *mut qtts_ctx,
        bytes: c_long,
        blocks: c_int,
        pattern: c_int,
        unroll: c_int,
        ld_bytes: c_int,
        iters: c_int,
        avg_us: *mut c_double,
    ) -> c_int;

    pub fn qtts_codec_decode(
        m: *mut qtts_model,
        codes: *const u32,
        lens: *const c_int,
        B: c_int,
        T_max: c_int,
        chunk: c_int,
        left_ctx: c_int,
        wav_out: *mut c_float,
        wav_stride: i64,
        wav_lens: *mut i64,
    ) -> c_int;
    pub fn qtts_codec_stream_open(m: *mut qtts_model, B: c_int, max_chunk_frames: c_int, out: *mut *mut qtts_codec_stream) -> c_int;
    pub fn qtts_codec_stream_push(
        s: *mut qtts_codec_stream,
        codes: *const u32,
        n_frames: c_int,
        wav_out: *mut c_float,
        wav_stride: i64,
        n_samples: *mut i64,
    ) -> c_int;
    pub fn qtts_codec_stream_close(s: *mut qtts_codec_stream);

    pub fn qtts_encode_audio(
        m: *mut qtts_model,
        audio: *const c_float,
        n_samples: *const c_int,
        B: c_int,
        max_samples: c_int,
        num_quantizers: c_int,
        codes_out: *mut u32,
        T_cap: c_int,
        T_out: *mut c_int,
    ) -> c_int;

    pub fn qtts_mel(
        m: *mut qtts_model,
        audio: *const c_float,
        n_samples: *const c_int,
        B: c_int,
        max_samples: c_int,
        mel_out: *mut c_float,
        frames_stride: c_int,
        n_frames: *mut c_int,
    ) -> c_int;
    pub fn qtts_mel_noalloc(
        m: *mut qtts_model,
        audio: *const c_float,
        n_samples: *const c_int,
        B: c_int,
        max_samples: c_int,
        mel_out: *mut c_float,
        frames_stride: c_int,
        n_frames: *mut c_int,
    ) -> c_int;
    pub fn qtts_mel_noalloc_capacity() -> i64;
    pub fn qtts_speaker_embed(
        m: *mut qtts_model,
        audio: *const c_float,
        n_samples: *const c_int,
        B: c_int,
        max_samples: c_int,
        emb_out: *mut c_float,
    ) -> c_int;
    pub fn qtts_speaker_embed_mel(
        m: *mut qtts_model,
        mel: *const c_float,
        n_frames: *const c_int,
        B: c_int,
        frames_stride: c_int,
        emb_out: *mut c_float,
    ) -> c_int;

    pub fn qtts_model_info(
        m: *mut qtts_model,
        hidden: *mut c_int,
        vocab: *mut c_int,
        groups: *mut c_int,
        total_upsample: *mut c_int,
        enc_dim: *mut c_int,
    ) -> c_int;
}
