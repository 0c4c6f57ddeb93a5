/* libqtts_hip -- C ABI of the MI355X (gfx950) Qwen3-TTS inference hot path.
 *
 * Drop-in boundary for the `qwen_tts` crate of tardigrade-dot/qwen3-tts-rs: a `hip` feature binds these
 * entry points (plain pointers and sizes, no C++ or torch types) in place of the Candle tensor calls
 * made by the functions cited on each declaration.  Citations are relative to the reference checkout
 * (qwen_tts/src/...).
 *
 * Conventions: every function returns 0 on success or a negative error class (QTTS_ERR_*), with a
 * message available from qtts_last_error() (thread local).  Nothing unwinds across the boundary.  The
 * caller owns every buffer it passes; the library owns the handles.  One qtts_ctx per GPU; a handle is
 * not thread-safe (callers serialise per handle, different GPUs may run from different threads).  All
 * host pointers are read / written synchronously before the call returns.
 */
#ifndef QTTS_H
#define QTTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define QTTS_ERR_BAD_ARG (-1)
#define QTTS_ERR_HIP (-2)
#define QTTS_ERR_OOM (-3)
#define QTTS_ERR_SHAPE (-4)
#define QTTS_ERR_STATE (-5)
#define QTTS_ERR_IO (-6)

#define QTTS_F32 0
#define QTTS_BF16 1

typedef struct qtts_ctx qtts_ctx;
typedef struct qtts_model qtts_model;

/* Device selection.  Replaces the Candle `Device` chosen in qwen_tts_cli/src/main.rs:138-161. */
int qtts_init(int device_ordinal, qtts_ctx** out);
void qtts_shutdown(qtts_ctx* ctx);
const char* qtts_last_error(void);
/* The HIP stream (hipStream_t) every kernel of this context is launched on; for event timing. */
void* qtts_stream(qtts_ctx* ctx);

/* Model construction.  Replaces ModelLoader::load_tts_model (model/loader.rs:267-324):
 * config_json is the text of config.json (config/tts_config.rs:15), codec_config_json the text of
 * speech_tokenizer/config.json (audio/tokenizer/v2/config.rs:9) or NULL for a model without codec.
 * dtype: QTTS_F32 or QTTS_BF16 (LoaderConfig.dtype, model/loader.rs:31-54). */
int qtts_model_create(qtts_ctx* ctx, const char* config_json, const char* codec_config_json, int dtype,
                      qtts_model** out);
/* One safetensors tensor, by its HF key (the VarBuilder paths of nn/talker.rs:32-59,253-277,
 * nn/code_predictor.rs:40-63,199-226, nn/speaker_encoder.rs:489-538, audio/decoder/v2.rs:53-134; codec
 * keys carry the "decoder." prefix of speech_tokenizer/model.safetensors).  data is host memory in
 * src_dtype (QTTS_F32 / QTTS_BF16); the library converts, uploads and later fuses (q|k|v, gate|up). */
int qtts_model_load_tensor(qtts_model* m, const char* hf_key, const void* data, int src_dtype, const int64_t* shape,
                           int rank);
/* Same, with `data` already resident in this GPU's memory (synthetic benchmarks; no reference counterpart). */
int qtts_model_load_tensor_device(qtts_model* m, const char* hf_key, const void* data, int src_dtype,
                                  const int64_t* shape, int rank);
/* Checks that every tensor the configs require is present, builds fused weights, RoPE tables and the
 * normalised codebooks (audio/quantizer.rs:56-59). */
int qtts_model_finalize(qtts_model* m);
/* The file side of ModelLoader::load_tts_model (model/loader.rs:267-324) in one call: <model_dir>/config.json,
 * speech_tokenizer/config.json (when load_tokenizer), every tensor of model.safetensors - or, as in the reference, of the FIRST
 * model-*.safetensors shard (model/loader.rs:219-235) - and of speech_tokenizer/model.safetensors through read-only mappings
 * (F32, BF16 and F16 tensors, others are skipped; encoder.* only when load_encoder, as with_encoder does), then qtts_model_finalize.
 * QTTS_ERR_IO: a file is missing or malformed. */
int qtts_model_load_dir(qtts_ctx* ctx, const char* model_dir, int dtype, int load_tokenizer, int load_encoder,
                        qtts_model** out);
void qtts_model_free(qtts_model* m);

/* Decode sessions.  No counterpart in the reference (one ConditionalGeneration serves one generate call at a time).
 * A session is a second decode stream on a finalized model: it borrows the model's weights (no copy) and owns a HIP stream,
 * KV caches, workspaces and the captured frame graph, so several batches can be in flight on one GPU while every weight byte
 * exists once.  The handle is a qtts_model handle: every entry point below that takes a qtts_model* (generate, codec decode
 * and streaming decode, mel, speaker embedding, audio encoding, timing, info) accepts it and runs on the session's stream;
 * qtts_model_load_tensor* and qtts_model_finalize fail on it.  Sessions of one model may be driven from different host
 * threads (one thread per session); the codec decoder's buffers belong to the handle too (codec decodes of several sessions run
 * side by side; up to QTTS_CODEC_SCRATCH_MB, default 24 GiB, each), while the speaker encoder and the codec encoder keep one set
 * of workspaces per model, so calls into each of them are serialised across sessions.  Close every session before
 * qtts_model_free of the model it was opened on. */
typedef qtts_model qtts_session;
int qtts_session_open(qtts_model* m, qtts_session** out);
void qtts_session_close(qtts_session* s);
/* The session's hipStream_t. */
void* qtts_session_stream(qtts_session* s);

/* One utterance.  Mirrors GenerationOptions (nn/generation_options.rs) + the text ids handed to
 * ConditionalGeneration::generate (nn/generation.rs:815). */
typedef struct {
    const uint32_t* text_ids;      /* <|im_start|>assistant\n ... <|im_end|>\n<|im_start|>assistant\n  */
    int n_text;
    const uint32_t* instruct_ids;  /* optional, NULL / 0 */
    int n_instruct;
    const char* language;          /* "auto", "english", ... (generation.rs:452-498) */
    const char* speaker;           /* name in talker_config.spk_id, or NULL */
    const float* speaker_embed;    /* optional x-vector, hidden_size floats (generation.rs:540-550) */
    const uint32_t* ref_codes;     /* optional ICL reference codes [ref_T][num_code_groups] */
    int ref_T;
    const uint32_t* ref_text_ids;  /* optional ICL reference text */
    int n_ref_text;
    int non_streaming;             /* GenerationOptions.non_streaming_mode (default 1) */
    int use_seed;                  /* 0: draws come from a time-seeded stream (nn/mt_rng.rs:204-214) */
    uint64_t seed;                 /* MT19937 seed of this utterance's draw stream (nn/mt_rng.rs:196) */
    int max_new_tokens;            /* qtts_generate_continuous only: this utterance's own frame limit (GenerationOptions.max_new_tokens
                                    * of a call for it alone); 0 = the call's max_new_tokens, larger values are clamped to it.  The
                                    * plain batch ignores it: there every row runs to the call's limit, as the reference's batch does. */
} qtts_request;

/* SamplingConfig (nn/sampling.rs:13-64) as merged by GenerateConfig::merge (model/config.rs:128-157).  The floating
 * point fields are f64 as in the reference and are narrowed exactly where it narrows them: (1 / temperature) as f32
 * (nn/sampling.rs:199-204), top_p as f32, repetition_penalty as f32.  min_new_tokens: frames during which EOS stays
 * suppressed (nn/sampling.rs:28-29,55).  A NEGATIVE value asks for the reference's default, 2 (what ConditionalGeneration::generate
 * sets: nn/generation.rs:841-842); 0 lets a row end on its first frame, and when every row does, num_steps is 0 as in
 * nn/generation.rs:939,1145-1150.  A zero-initialised struct therefore does NOT give the reference's behaviour: set the
 * field to -1 (or 2) for that. */
typedef struct {
    double temperature;
    int top_k;
    double top_p;
    int do_sample;
    double repetition_penalty;
    int subtalker_do_sample;
    double subtalker_temperature;
    int subtalker_top_k;
    double subtalker_top_p;
    int min_new_tokens;
} qtts_sampling;

/* Output{codes, num_steps, effective_lengths} (nn/generation.rs:31-38) for one utterance.
 * codes: caller-allocated max_new_tokens * num_code_groups u32, filled as [num_steps][groups]. */
typedef struct {
    uint32_t* codes;
    int num_steps;
    int effective_len;
} qtts_codes_out;

/* Optional per-step traces for parity tests (NULL pointers are skipped); each is caller-allocated for
 * max_new_tokens steps of utterance 0..n-1: logits [n][max_new][vocab], hidden [n][max_new][hidden]. */
typedef struct {
    float* logits;
    float* hidden;
    int use_graph;     /* 1: replay the decode frame as a hipGraph (default path), 0: eager launches */
    int profile_gemm;  /* eager only: time every weight-streaming GEMM launch with HIP events */
    /* First-audio hook (no counterpart in the reference, whose generate returns only when every frame exists): once
     * `first_frames` frames of every row exist, chunk_cb(chunk_user, codes [n][first_frames][groups], n, first_frames) is
     * called on the calling thread (at the end of generation with fewer frames if it stops earlier).  The codec decoder
     * is causal, so decoding that prefix gives final audio for all but its last two frames; the callback may call
     * qtts_codec_decode on the same model. */
    int first_frames;
    void (*chunk_cb)(void* user, const uint32_t* codes, int n, int frames);
    void* chunk_user;
    /* Throughput-measurement protocol (no counterpart in the reference): when > 0, overrides qtts_sampling.min_new_tokens so
     * that every row of a benchmark run produces the same fixed length. */
    int min_new_tokens;
    /* Optional trace of the code predictor's logits (eager launches only, like logits / hidden):
     * [n][max_new][num_code_groups - 1][code predictor vocab]. */
    float* cp_logits;
    /* 1: the rows of a qtts_generate(_ex) batch draw from ONE MT19937 stream, seeded with reqs[0].seed, in the order of the
     * reference's global generator (nn/mt_rng.rs:22): per frame the talker draws of rows 0 .. n-1 (nn/generation.rs:1055-1060),
     * then for each residual codebook the draws of rows 0 .. n-1 (nn/code_predictor.rs:468-472,517-522); a greedy stage draws
     * nothing.  A caller that hands the reference's own batch to one call gets the reference's tokens.  0 (default): every
     * request owns the stream of its seed, so a row's tokens do not depend on its batch.  Not available with continuous
     * batching (the draw order of a slot schedule has no counterpart in the reference). */
    int shared_rng_stream;
    /* qtts_generate_continuous only: done_cb(done_user, i, codes [num_steps][groups], num_steps, effective_len) is called on the
     * calling thread as soon as utterance i has left its row (the values outs[i] receives; codes is valid during the call only).
     * Between frames the handle's stream is idle, so the callback may run qtts_codec_decode on the same handle - finished
     * utterances become audio while the other rows keep decoding.  It must not call qtts_generate* on that handle. */
    void (*done_cb)(void* user, int utterance, const uint32_t* codes, int num_steps, int effective_len);
    void* done_user;
} qtts_gen_debug;

/* ConditionalGeneration::generate (nn/generation.rs:815-1194) for n independent utterances decoded as
 * one batch; row i is token-identical to a batch-1 call with reqs[i] (rows never interact). */
int qtts_generate(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens,
                  qtts_codes_out* outs);
int qtts_generate_ex(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens,
                     qtts_codes_out* outs, const qtts_gen_debug* dbg);

/* Continuous batching (no counterpart in the reference, whose batch runs until its last row is done): n utterances are decoded
 * through at most `slots` rows.  A row is retired once it has produced the frame with its EOS or its frame limit (the call's
 * max_new_tokens, or the request's own when it sets one), and its slot is prefilled with the next queued utterance while the
 * other rows keep decoding (rows are polled every 8 frames).  Every utterance gets the computation a qtts_generate call for it
 * alone performs: outs[i].num_steps = its own frame count (the EOS frame included, 0 if its first token is EOS), effective_len as
 * there; with a seeded request the draws are the same stream.  Rows never interact, and admission work (text projection, prompt
 * prefill) runs through one kernel per GEMM shape whatever the number of prompts admitted together, so an utterance's codes are
 * bit for bit those of the same request in a qtts_generate batch of the same row count (`slots` rows): the decode kernels are
 * picked by launch width, and widths of different kernel classes (up to 16, 32, 64, 128 ... rows) sum in different orders, which
 * in bf16 moves sampled tokens.  In f32 at the shapes of the tests the result equals the batch of one.
 * outs[i].codes holds max_new_tokens * groups u32 as for qtts_generate.  dbg may carry use_graph, profile_gemm, min_new_tokens and
 * the done_cb hook; traces and the first-chunk hook are not available here. */
int qtts_generate_continuous(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens,
                             int slots, qtts_codes_out* outs, const qtts_gen_debug* dbg);

/* Timing of the last qtts_generate_ex call (milliseconds, HIP events on the context stream):
 * out[0] prompt+prefill, out[1] frame loop, out[2] summed GEMM kernel time (profile_gemm only),
 * out[3] GEMM launches timed, out[4] algorithmic weight bytes those launches streamed, out[5] frames run,
 * out[6] device time of the last qtts_codec_decode call excluding the waveform copy to the host. */
int qtts_last_timing(qtts_model* m, double* out, int n);

/* Measurement hooks (no counterpart in the reference; used by the benchmark to report rooflines).
 * qtts_gemm_profile: per-shape totals of the last profile_gemm run, rows of 5 doubles
 *   (M tokens, N, K, launches, total microseconds); *n_rows receives the number of shapes.
 * qtts_bench_gemm: average duration (us, HIP events around `iters` back-to-back launches on the context stream)
 *   of the weight-streaming GEMM out[M][N] = X[M][K] W[N][K]^T, rotating over weight copies so that launches do
 *   not hit the Infinity Cache; KS <= 0 picks the split-K factor the decode path would use.
 * qtts_bench_stream: same for a pure read of `bytes` bytes (pattern 0 contiguous, 1 row-tiled). */
int qtts_gemm_profile(qtts_model* m, double* rows, int cap_rows, int* n_rows);
int qtts_bench_gemm(qtts_ctx* ctx, int dtype, int M, int N, int K, int KS, int iters, double* avg_us);
int qtts_bench_stream(qtts_ctx* ctx, long bytes, int blocks, int pattern, int unroll, int ld_bytes, int iters, double* avg_us);

/* Model::decode_output / TokenizerV2::{decode,chunked_decode} (model/generate.rs:29-143,
 * audio/tokenizer/v2/mod.rs:208-232, audio/decoder/v2.rs:163-286).  codes [B][T_max][Q]; lens[b] frames
 * of row b are decoded; chunk <= 0 decodes whole sequences, chunk > 0 uses the reference's chunked
 * decode (windows of `chunk` frames with `left_ctx` frames of left context), chunk == -2 applies the
 * decode_output policy (> 512 frames => 256/32).  wav_out [B][wav_stride] floats; wav_lens[b] receives
 * the number of samples produced for row b. */
int qtts_codec_decode(qtts_model* m, const uint32_t* codes, const int* lens, int B, int T_max, int chunk, int left_ctx,
                      float* wav_out, int64_t wav_stride, int64_t* wav_lens);

/* Streaming decode session.  No counterpart in the reference, whose only incremental mode is chunked_decode
 * (audio/decoder/v2.rs:250-286: every chunk restarts with 32 frames of context, so its output differs from a whole decode).
 * Here the concatenation of the pushes' samples EQUALS qtts_codec_decode(chunk = 0) of the concatenated codes: the
 * transformer continues from a per-row ring of keys/values and the conv stack re-decodes the few context frames its
 * receptive field spans.  All B rows advance together.  A push of n frames (1 <= n <= max_chunk_frames), codes [B][n][Q],
 * writes to wav_out [B][wav_stride] the samples those frames decide: up * n - tail for the first push (1920 n - 555 at the
 * default config), up * n for every later one; *n_samples receives that count. */
typedef struct qtts_codec_stream qtts_codec_stream;
int qtts_codec_stream_open(qtts_model* m, int B, int max_chunk_frames, qtts_codec_stream** out);
int qtts_codec_stream_push(qtts_codec_stream* s, const uint32_t* codes, int n_frames, float* wav_out, int64_t wav_stride,
                           int64_t* n_samples);
void qtts_codec_stream_close(qtts_codec_stream* s);

/* TokenizerV2::encode / Model::encode_audio (audio/tokenizer/v2/mod.rs:164-171, audio/encoder/v2.rs:87-120): 24 kHz clips
 * audio [B][max_samples] -> codes_out [B][T_cap][num_quantizers], T_out[b] = n_samples[b] / 1920 frames.  Needs the speech
 * tokenizer's `encoder.*` tensors (acoustic codebooks beyond num_quantizers - 1 may be omitted). */
int qtts_encode_audio(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, int num_quantizers,
                      uint32_t* codes_out, int T_cap, int* T_out);

/* mel_spectrogram (audio/mel.rs:81-158, default config) as used by Model::compute_mel_spectrogram
 * (model/mod.rs:520-528): audio [B][max_samples] f32 @ 24 kHz -> mel_out [B][frames_stride][128]. */
int qtts_mel(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out,
             int frames_stride, int* n_frames);
/* qtts_mel for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc, processes
 * whose other streams are being captured into HIP graphs: an allocation synchronises the device; the call itself copies from
 * pageable host memory and waits for its stream, so it is not meant to be captured).  No counterpart in the reference.  Same arguments and results as qtts_mel, but the device-side staging
 * of the clips, their lengths and the result lives in one fixed array per GPU (qtts_mel_noalloc_capacity() bytes, 16 MiB)
 * instead of buffers grown on demand: the call issues no hipMalloc / hipFree.  It fails with QTTS_ERR_SHAPE when
 * pad256(4 B) + pad256(4 B max_samples) + 512 B frames_stride exceeds that capacity.  Calls on one GPU share the array: the
 * library makes them take turns (one lock per GPU, whatever model or session they come through). */
int qtts_mel_noalloc(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out,
                     int frames_stride, int* n_frames);
int64_t qtts_mel_noalloc_capacity(void);
/* ConditionalGeneration::encode_speaker (nn/generation.rs:140) on top of qtts_mel: emb_out [B][enc_dim]. */
int qtts_speaker_embed(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* emb_out);
/* The same on a mel the caller already holds - Model::create_voice_clone_prompt_from_mel (model/mod.rs:579-612):
 * mel [B][frames_stride][128] f32 log-mel (frames x mels, the layout qtts_mel writes), n_frames[b] valid frames. */
int qtts_speaker_embed_mel(qtts_model* m, const float* mel, const int* n_frames, int B, int frames_stride, float* emb_out);

/* Shape queries a binding needs to size its buffers. */
int qtts_model_info(qtts_model* m, int* hidden, int* vocab, int* groups, int* total_upsample, int* enc_dim);

#ifdef __cplusplus
}
#endif
#endif /* QTTS_H */
