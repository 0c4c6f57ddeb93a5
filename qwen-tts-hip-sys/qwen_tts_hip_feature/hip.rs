//! `qwen_tts/src/hip.rs` — the `hip` feature's backend: the hot path of the crate behind `libqtts_hip.so`.
//!
//! This file is new in the `qwen_tts` crate (it replaces nothing); the three edits to existing files are listed in
//! README.md next to it.  It owns the handles, converts the crate's option structs to the C structs, and returns plain
//! vectors — the callers in `model/` wrap them into `Output` / waveforms exactly where the Candle path produces tensors.
//! Not compiled in the image this repository is built in (no Rust toolchain there).
use std::ffi::{CStr, CString};
use std::os::raw::{c_int, c_void};
use std::ptr;

use qwen_tts_hip_sys as sys;

#[derive(Debug)]
pub struct HipError {
    pub code: i32,
    pub message: String,
}
impl std::fmt::Display for HipError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "libqtts_hip error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for HipError {}

fn check(rc: c_int) -> Result<(), HipError> {
    if rc == 0 {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(sys::qtts_last_error()) }.to_string_lossy().into_owned();
    Err(HipError { code: rc, message })
}

/// One GPU.  `qtts_init` + `qtts_shutdown`.
pub struct HipDevice {
    ctx: *mut sys::qtts_ctx,
}
unsafe impl Send for HipDevice {}
impl HipDevice {
    pub fn new(ordinal: usize) -> Result<Self, HipError> {
        let mut ctx = ptr::null_mut();
        check(unsafe { sys::qtts_init(ordinal as c_int, &mut ctx) })?;
        Ok(Self { ctx })
    }
}
impl Drop for HipDevice {
    fn drop(&mut self) {
        unsafe { sys::qtts_shutdown(self.ctx) }
    }
}

/// Sampling parameters as `SamplingConfig` holds them after `GenerateConfig::merge`.
#[derive(Clone, Copy, Debug)]
pub struct HipSampling {
    pub temperature: f64,
    pub top_k: usize,
    pub top_p: f64,
    pub do_sample: bool,
    pub repetition_penalty: f64,
    pub min_new_tokens: usize,
    pub subtalker_do_sample: bool,
    pub subtalker_temperature: f64,
    pub subtalker_top_k: usize,
    pub subtalker_top_p: f64,
}
impl HipSampling {
    fn to_c(self) -> sys::qtts_sampling {
        sys::qtts_sampling {
            temperature: self.temperature,
            top_k: self.top_k as c_int,
            top_p: self.top_p,
            do_sample: self.do_sample as c_int,
            repetition_penalty: self.repetition_penalty,
            subtalker_do_sample: self.subtalker_do_sample as c_int,
            subtalker_temperature: self.subtalker_temperature,
            subtalker_top_k: self.subtalker_top_k as c_int,
            subtalker_top_p: self.subtalker_top_p,
            min_new_tokens: self.min_new_tokens as c_int,
        }
    }
}

/// One utterance: the text ids handed to `ConditionalGeneration::generate` plus the fields of `GenerationOptions`.
#[derive(Default, Clone)]
pub struct HipRequest {
    pub text_ids: Vec<u32>,
    pub instruct_ids: Vec<u32>,
    pub language: String,
    pub speaker: Option<String>,
    pub speaker_embed: Option<Vec<f32>>,
    /// ICL reference codes, row-major `[ref_t][num_code_groups]`
    pub ref_codes: Option<(Vec<u32>, usize)>,
    pub ref_text_ids: Vec<u32>,
    pub non_streaming_mode: bool,
    pub seed: Option<u64>,
}

/// What `Output` carries, per utterance.
pub struct HipCodes {
    /// `[num_steps][num_code_groups]`
    pub codes: Vec<u32>,
    pub num_steps: usize,
    pub effective_len: usize,
}

/// A loaded model, or a decode session opened on one (`open_session`): both are `qtts_model` handles.
pub struct HipModel {
    m: *mut sys::qtts_model,
    session: bool,
    pub hidden: usize,
    pub vocab: usize,
    pub groups: usize,
    pub total_upsample: usize,
    pub enc_dim: usize,
}
unsafe impl Send for HipModel {}

impl HipModel {
    /// `ModelLoader::load_tts_model`: `config_json` / `codec_json` are the texts of config.json and
    /// speech_tokenizer/config.json; `tensors` yields every safetensors tensor as (HF key, bytes, is_bf16, shape).
    pub fn load<'a, I>(dev: &HipDevice, config_json: &str, codec_json: Option<&str>, bf16: bool, tensors: I) -> Result<Self, HipError>
    where
        I: IntoIterator<Item = (&'a str, &'a [u8], bool, &'a [i64])>,
    {
        let cfg = CString::new(config_json).expect("config.json holds no NUL");
        let codec = codec_json.map(|s| CString::new(s).expect("codec config holds no NUL"));
        let mut m = ptr::null_mut();
        check(unsafe {
            sys::qtts_model_create(
                dev.ctx,
                cfg.as_ptr(),
                codec.as_ref().map_or(ptr::null(), |c| c.as_ptr()),
                if bf16 { sys::QTTS_BF16 } else { sys::QTTS_F32 },
                &mut m,
            )
        })?;
        let mut model = Self { m, session: false, hidden: 0, vocab: 0, groups: 0, total_upsample: 0, enc_dim: 0 };
        for (key, bytes, is_bf16, shape) in tensors {
            let k = CString::new(key).expect("tensor key holds no NUL");
            check(unsafe {
                sys::qtts_model_load_tensor(
                    model.m,
                    k.as_ptr(),
                    bytes.as_ptr() as *const c_void,
                    if is_bf16 { sys::QTTS_BF16 } else { sys::QTTS_F32 },
                    shape.as_ptr(),
                    shape.len() as c_int,
                )
            })?;
        }
        check(unsafe { sys::qtts_model_finalize(model.m) })?;
        model.query_info()?;
        Ok(model)
    }

    /// The same from a checkpoint directory, read by the library itself (`qtts_model_load_dir`): config.json,
    /// speech_tokenizer/config.json and the safetensors files, mapped read-only; `with_encoder` as in `ModelLoader::with_encoder`.
    pub fn load_dir(dev: &HipDevice, model_dir: &std::path::Path, bf16: bool, load_tokenizer: bool, with_encoder: bool) -> Result<Self, HipError> {
        let dir = CString::new(model_dir.to_string_lossy().as_bytes()).expect("path holds no NUL");
        let mut m = ptr::null_mut();
        check(unsafe {
            sys::qtts_model_load_dir(
                dev.ctx,
                dir.as_ptr(),
                if bf16 { sys::QTTS_BF16 } else { sys::QTTS_F32 },
                load_tokenizer as c_int,
                with_encoder as c_int,
                &mut m,
            )
        })?;
        let mut model = Self { m, session: false, hidden: 0, vocab: 0, groups: 0, total_upsample: 0, enc_dim: 0 };
        model.query_info()?;
        Ok(model)
    }

    fn query_info(&mut self) -> Result<(), HipError> {
        let (mut h, mut v, mut g, mut u, mut e) = (0, 0, 0, 0, 0);
        check(unsafe { sys::qtts_model_info(self.m, &mut h, &mut v, &mut g, &mut u, &mut e) })?;
        self.hidden = h as usize;
        self.vocab = v as usize;
        self.groups = g as usize;
        self.total_upsample = u as usize;
        self.enc_dim = e as usize;
        Ok(())
    }

    /// A second decode stream on the same weights; drop it before the model it was opened on.
    pub fn open_session(&self) -> Result<HipModel, HipError> {
        let mut s = ptr::null_mut();
        check(unsafe { sys::qtts_session_open(self.m, &mut s) })?;
        let mut ses = HipModel { m: s, session: true, hidden: 0, vocab: 0, groups: 0, total_upsample: 0, enc_dim: 0 };
        ses.query_info()?;
        Ok(ses)
    }

    /// `ConditionalGeneration::generate` for a batch (`generate_custom_voice_from_texts_batch` makes one call of it).
    pub fn generate(&mut self, reqs: &[HipRequest], sampling: HipSampling, max_new_tokens: usize) -> Result<Vec<HipCodes>, HipError> {
        let langs: Vec<CString> = reqs.iter().map(|r| CString::new(r.language.as_str()).unwrap()).collect();
        let spks: Vec<Option<CString>> = reqs.iter().map(|r| r.speaker.as_ref().map(|s| CString::new(s.as_str()).unwrap())).collect();
        let c_reqs: Vec<sys::qtts_request> = reqs
            .iter()
            .enumerate()
            .map(|(i, r)| sys::qtts_request {
                text_ids: r.text_ids.as_ptr(),
                n_text: r.text_ids.len() as c_int,
                instruct_ids: if r.instruct_ids.is_empty() { ptr::null() } else { r.instruct_ids.as_ptr() },
                n_instruct: r.instruct_ids.len() as c_int,
                language: langs[i].as_ptr(),
                speaker: spks[i].as_ref().map_or(ptr::null(), |s| s.as_ptr()),
                speaker_embed: r.speaker_embed.as_ref().map_or(ptr::null(), |e| e.as_ptr()),
                ref_codes: r.ref_codes.as_ref().map_or(ptr::null(), |(c, _)| c.as_ptr()),
                ref_T: r.ref_codes.as_ref().map_or(0, |(_, t)| *t as c_int),
                ref_text_ids: if r.ref_text_ids.is_empty() { ptr::null() } else { r.ref_text_ids.as_ptr() },
                n_ref_text: r.ref_text_ids.len() as c_int,
                non_streaming: r.non_streaming_mode as c_int,
                use_seed: r.seed.is_some() as c_int,
                seed: r.seed.unwrap_or(0),
                max_new_tokens: 0,
            })
            .collect();
        let mut bufs: Vec<Vec<u32>> = reqs.iter().map(|_| vec![0u32; max_new_tokens * self.groups]).collect();
        let mut outs: Vec<sys::qtts_codes_out> =
            bufs.iter_mut().map(|b| sys::qtts_codes_out { codes: b.as_mut_ptr(), num_steps: 0, effective_len: 0 }).collect();
        let s = sampling.to_c();
        check(unsafe { sys::qtts_generate(self.m, c_reqs.as_ptr(), c_reqs.len() as c_int, &s, max_new_tokens as c_int, outs.as_mut_ptr()) })?;
        Ok(outs
            .iter()
            .zip(bufs)
            .map(|(o, mut b)| {
                b.truncate(o.num_steps as usize * self.groups);
                HipCodes { codes: b, num_steps: o.num_steps as usize, effective_len: o.effective_len as usize }
            })
            .collect())
    }

    /// `Model::decode_output` (chunk = -2 applies its 512-frame policy): codes `[b][t_max][q]`, one waveform per row.
    pub fn decode(&mut self, codes: &[u32], lens: &[usize], t_max: usize, chunk: i32, left_ctx: i32) -> Result<Vec<Vec<f32>>, HipError> {
        let b = lens.len();
        let stride = t_max * self.total_upsample;
        let mut wav = vec![0f32; b * stride];
        let mut wav_lens = vec![0i64; b];
        let l: Vec<c_int> = lens.iter().map(|&x| x as c_int).collect();
        check(unsafe {
            sys::qtts_codec_decode(self.m, codes.as_ptr(), l.as_ptr(), b as c_int, t_max as c_int, chunk, left_ctx, wav.as_mut_ptr(), stride as i64, wav_lens.as_mut_ptr())
        })?;
        Ok((0..b).map(|i| wav[i * stride..i * stride + wav_lens[i] as usize].to_vec()).collect())
    }

    /// `TokenizerV2::encode`: 24 kHz clips `[b][max_samples]` -> (codes `[b][t_cap][q]`, frames per row).
    pub fn encode_audio(&mut self, audio: &[f32], n_samples: &[usize], max_samples: usize, q: usize) -> Result<(Vec<u32>, Vec<usize>, usize), HipError> {
        let b = n_samples.len();
        let t_cap = max_samples / self.total_upsample + 1;
        let mut codes = vec![0u32; b * t_cap * q];
        let mut t_out = vec![0 as c_int; b];
        let n: Vec<c_int> = n_samples.iter().map(|&x| x as c_int).collect();
        check(unsafe {
            sys::qtts_encode_audio(self.m, audio.as_ptr(), n.as_ptr(), b as c_int, max_samples as c_int, q as c_int, codes.as_mut_ptr(), t_cap as c_int, t_out.as_mut_ptr())
        })?;
        Ok((codes, t_out.iter().map(|&t| t as usize).collect(), t_cap))
    }

    /// `Model::compute_mel_spectrogram`: (mel `[b][frames_stride][128]`, frames per row).
    pub fn mel(&mut self, audio: &[f32], n_samples: &[usize], max_samples: usize) -> Result<(Vec<f32>, Vec<usize>, usize), HipError> {
        let b = n_samples.len();
        let frames_stride = max_samples / 256 + 2;
        let mut mel = vec![0f32; b * frames_stride * 128];
        let mut frames = vec![0 as c_int; b];
        let n: Vec<c_int> = n_samples.iter().map(|&x| x as c_int).collect();
        check(unsafe { sys::qtts_mel(self.m, audio.as_ptr(), n.as_ptr(), b as c_int, max_samples as c_int, mel.as_mut_ptr(), frames_stride as c_int, frames.as_mut_ptr()) })?;
        Ok((mel, frames.iter().map(|&f| f as usize).collect(), frames_stride))
    }

    /// `ConditionalGeneration::encode_speaker`: one x-vector of `enc_dim` floats per clip.
    pub fn speaker_embed(&mut self, audio: &[f32], n_samples: &[usize], max_samples: usize) -> Result<Vec<f32>, HipError> {
        let b = n_samples.len();
        let mut emb = vec![0f32; b * self.enc_dim];
        let n: Vec<c_int> = n_samples.iter().map(|&x| x as c_int).collect();
        check(unsafe { sys::qtts_speaker_embed(self.m, audio.as_ptr(), n.as_ptr(), b as c_int, max_samples as c_int, emb.as_mut_ptr()) })?;
        Ok(emb)
    }

    /// `Model::create_voice_clone_prompt_from_mel`: x-vectors from mels the caller holds (`[b][frames_stride][128]`).
    pub fn speaker_embed_mel(&mut self, mel: &[f32], n_frames: &[usize], frames_stride: usize) -> Result<Vec<f32>, HipError> {
        let b = n_frames.len();
        let mut emb = vec![0f32; b * self.enc_dim];
        let n: Vec<c_int> = n_frames.iter().map(|&x| x as c_int).collect();
        check(unsafe { sys::qtts_speaker_embed_mel(self.m, mel.as_ptr(), n.as_ptr(), b as c_int, frames_stride as c_int, emb.as_mut_ptr()) })?;
        Ok(emb)
    }
}

impl Drop for HipModel {
    fn drop(&mut self) {
        unsafe {
            if self.session {
                sys::qtts_session_close(self.m)
            } else {
                sys::qtts_model_free(self.m)
            }
        }
    }
}

/// Streaming codec decode (no counterpart in the reference): the pushes' samples concatenated equal a whole decode.
pub struct HipCodecStream {
    s: *mut sys::qtts_codec_stream,
    rows: usize,
    max_chunk: usize,
    upsample: usize,
}
impl HipCodecStream {
    pub fn open(model: &mut HipModel, rows: usize, max_chunk_frames: usize) -> Result<Self, HipError> {
        let mut s = ptr::null_mut();
        check(unsafe { sys::qtts_codec_stream_open(model.m, rows as c_int, max_chunk_frames as c_int, &mut s) })?;
        Ok(Self { s, rows, max_chunk: max_chunk_frames, upsample: model.total_upsample })
    }
    /// codes `[rows][n_frames][q]` -> samples `[rows][n]` those frames decide
    pub fn push(&mut self, codes: &[u32], n_frames: usize) -> Result<Vec<Vec<f32>>, HipError> {
        assert!(n_frames >= 1 && n_frames <= self.max_chunk);
        let stride = n_frames * self.upsample;
        let mut wav = vec![0f32; self.rows * stride];
        let mut n = 0i64;
        check(unsafe { sys::qtts_codec_stream_push(self.s, codes.as_ptr(), n_frames as c_int, wav.as_mut_ptr(), stride as i64, &mut n) })?;
        Ok((0..self.rows).map(|r| wav[r * stride..r * stride + n as usize].to_vec()).collect())
    }
}
impl Drop for HipCodecStream {
    fn drop(&mut self) {
        unsafe { sys::qtts_codec_stream_close(self.s) }
    }
}
