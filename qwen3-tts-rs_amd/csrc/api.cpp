// C ABI of libqtts_hip (declared in include/qtts.h).  Exceptions never cross the boundary: they are
// mapped to the negative error classes and a thread-local message.
#include <atomic>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "generate.hpp"
#include "loader.hpp"

using namespace qtts;

struct qtts_ctx {
    Ctx c;
};
struct qtts_model {
    Model m;
    std::unique_ptr<Ctx> own_ctx;  // sessions: their own stream
};

namespace {
thread_local std::string g_err;
int fail(const std::exception& e, int code) {
    g_err = e.what();
    return code;
}
}  // namespace

#define QTTS_API_BEGIN try {
#define QTTS_API_END                                  \
    return 0;                                         \
    }                                                 \
    catch (const qtts::Error& e) {                    \
        return fail(e, e.code);                       \
    }                                                 \
    catch (const std::bad_alloc& e) {                 \
        return fail(e, ERR_OOM);                      \
    }                                                 \
    catch (const std::exception& e) {                 \
        return fail(e, ERR_STATE);                    \
    }                                                 \
    catch (...) {                                     \
        g_err = "unknown exception";                  \
        return ERR_STATE;                             \
    }

extern "C" {

const char* qtts_last_error(void) { return g_err.c_str(); }

// Experiment switch QTTS_STREAM_PARTS=n (2..8): stream number k of the process (context streams and session streams in creation
// order) is confined to the compute units whose index is k modulo n, so that n streams never queue for each other's CUs.
static hipError_t create_stream(hipStream_t* st) {
    static const int parts = getenv("QTTS_STREAM_PARTS") ? atoi(getenv("QTTS_STREAM_PARTS")) : 0;
    static std::atomic<int> created{0};
    if (parts < 2 || parts > 8) return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    const int k = created.fetch_add(1) % parts;
    hipDeviceProp_t prop;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((size_t)(ncu + 31) / 32, 0u);
    for (int i = 0; i < ncu; ++i)
        if (i % parts == k) mask[(size_t)i / 32] |= 1u << (i % 32);
    return hipExtStreamCreateWithCUMask(st, (uint32_t)mask.size(), mask.data());
}

int qtts_init(int device_ordinal, qtts_ctx** out) {
    QTTS_API_BEGIN
    QTTS_CHECK(out != nullptr, ERR_BAD_ARG, "qtts_init: null out");
    int count = 0;
    QTTS_HIP(hipGetDeviceCount(&count));
    QTTS_CHECK(device_ordinal >= 0 && device_ordinal < count, ERR_BAD_ARG, "qtts_init: no such device");
    QTTS_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    QTTS_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    QTTS_CHECK(std::string(prop.gcnArchName).rfind("gfx950", 0) == 0, ERR_STATE,
               std::string("qtts_init: kernels are built for gfx950 only, found ") + prop.gcnArchName);
    std::unique_ptr<qtts_ctx> c(new qtts_ctx());
    c->c.device = device_ordinal;
    QTTS_HIP(create_stream(&c->c.stream));
    try {
        (void)mel_scratch(device_ordinal);  // device address of this GPU's fixed staging array (qtts_mel_noalloc)
    } catch (...) {
        (void)hipStreamDestroy(c->c.stream);
        throw;
    }
    *out = c.release();
    QTTS_API_END
}

void qtts_shutdown(qtts_ctx* ctx) {
    if (!ctx) return;
    if (ctx->c.stream) (void)hipStreamDestroy(ctx->c.stream);
    delete ctx;
}

void* qtts_stream(qtts_ctx* ctx) { return ctx ? (void*)ctx->c.stream : nullptr; }

int qtts_model_create(qtts_ctx* ctx, const char* config_json, const char* codec_config_json, int dtype, qtts_model** out) {
    QTTS_API_BEGIN
    QTTS_CHECK(ctx && config_json && out, ERR_BAD_ARG, "qtts_model_create: null argument");
    QTTS_CHECK(dtype == F32 || dtype == BF16, ERR_BAD_ARG, "qtts_model_create: dtype must be QTTS_F32 or QTTS_BF16");
    std::unique_ptr<qtts_model> m(new qtts_model());
    m->m.ctx = &ctx->c;
    m->m.dtype = dtype;
    m->m.parse_config(config_json, codec_config_json);
    *out = m.release();
    QTTS_API_END
}

int qtts_model_load_tensor(qtts_model* m, const char* hf_key, const void* data, int src_dtype, const int64_t* shape, int rank) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && hf_key && shape, ERR_BAD_ARG, "qtts_model_load_tensor: null argument");
    m->m.load_tensor(hf_key, data, src_dtype, shape, rank, false);
    QTTS_API_END
}

int qtts_model_load_tensor_device(qtts_model* m, const char* hf_key, const void* data, int src_dtype, const int64_t* shape,
                                  int rank) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && hf_key && shape, ERR_BAD_ARG, "qtts_model_load_tensor_device: null argument");
    m->m.load_tensor(hf_key, data, src_dtype, shape, rank, true);
    QTTS_API_END
}

int qtts_model_finalize(qtts_model* m) {
    QTTS_API_BEGIN
    QTTS_CHECK(m, ERR_BAD_ARG, "qtts_model_finalize: null model");
    m->m.finalize();
    QTTS_API_END
}

int qtts_model_load_dir(qtts_ctx* ctx, const char* model_dir, int dtype, int load_tokenizer, int load_encoder, qtts_model** out) {
    QTTS_API_BEGIN
    QTTS_CHECK(ctx && model_dir && out, ERR_BAD_ARG, "qtts_model_load_dir: null argument");
    QTTS_CHECK(dtype == F32 || dtype == BF16, ERR_BAD_ARG, "qtts_model_load_dir: dtype must be QTTS_F32 or QTTS_BF16");
    std::unique_ptr<qtts_model> m(new qtts_model());
    m->m.ctx = &ctx->c;
    m->m.dtype = dtype;
    QTTS_HIP(hipSetDevice(ctx->c.device));
    load_model_dir(m->m, model_dir, load_tokenizer != 0, load_encoder != 0);
    *out = m.release();
    QTTS_API_END
}

void qtts_model_free(qtts_model* m) {
    if (!m) return;
    (void)hipSetDevice(m->m.ctx->device);
    (void)hipDeviceSynchronize();
    hipStream_t own = m->own_ctx ? m->own_ctx->stream : nullptr;
    m->m.gen.reset();  // graphs and workspaces go before the stream they were captured on
    if (own) (void)hipStreamDestroy(own);
    delete m;
}

int qtts_session_open(qtts_model* base, qtts_model** out) {
    QTTS_API_BEGIN
    QTTS_CHECK(base && out, ERR_BAD_ARG, "qtts_session_open: null argument");
    QTTS_CHECK(base->m.finalized, ERR_STATE, "qtts_session_open: the model is not finalized");
    std::unique_ptr<qtts_model> s(new qtts_model());
    s->own_ctx.reset(new Ctx());
    s->own_ctx->device = base->m.ctx->device;
    QTTS_HIP(hipSetDevice(s->own_ctx->device));
    QTTS_HIP(create_stream(&s->own_ctx->stream));
    s->m.ctx = s->own_ctx.get();
    try {
        s->m.init_view(base->m);
    } catch (...) {
        (void)hipStreamDestroy(s->own_ctx->stream);
        throw;
    }
    *out = s.release();
    QTTS_API_END
}

void qtts_session_close(qtts_model* s) { qtts_model_free(s); }

void* qtts_session_stream(qtts_model* s) { return s ? (void*)s->m.ctx->stream : nullptr; }

int qtts_generate_ex(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens,
                     qtts_codes_out* outs, const qtts_gen_debug* dbg) {
    QTTS_API_BEGIN
    QTTS_CHECK(m, ERR_BAD_ARG, "qtts_generate: null model");
    run_generate(m->m, reqs, n, s, max_new_tokens, outs, dbg);
    QTTS_API_END
}

int qtts_generate(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens,
                  qtts_codes_out* outs) {
    return qtts_generate_ex(m, reqs, n, s, max_new_tokens, outs, nullptr);
}

int qtts_generate_continuous(qtts_model* m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens, int slots,
                             qtts_codes_out* outs, const qtts_gen_debug* dbg) {
    QTTS_API_BEGIN
    QTTS_CHECK(m, ERR_BAD_ARG, "qtts_generate_continuous: null model");
    run_generate_continuous(m->m, reqs, n, s, max_new_tokens, slots, outs, dbg);
    QTTS_API_END
}

int qtts_last_timing(qtts_model* m, double* out, int n) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && out && n >= 0, ERR_BAD_ARG, "qtts_last_timing: null argument");
    for (int i = 0; i < n && i < 8; ++i) out[i] = m->m.timing[i];
    QTTS_API_END
}

int qtts_gemm_profile(qtts_model* m, double* rows, int cap_rows, int* n_rows) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && rows && n_rows && cap_rows >= 0, ERR_BAD_ARG, "qtts_gemm_profile: null argument");
    const int n = (int)(m->m.gemm_profile.size() / 5);
    *n_rows = n;
    for (int i = 0; i < n && i < cap_rows; ++i)
        for (int j = 0; j < 5; ++j) rows[i * 5 + j] = m->m.gemm_profile[(size_t)i * 5 + j];
    QTTS_API_END
}

int qtts_codec_decode(qtts_model* m, const uint32_t* codes, const int* lens, int B, int T_max, int chunk, int left_ctx,
                      float* wav_out, int64_t wav_stride, int64_t* wav_lens) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && codes && lens && wav_out && wav_lens, ERR_BAD_ARG, "qtts_codec_decode: null argument");
    run_codec_decode(m->m, codes, lens, B, T_max, chunk, left_ctx, wav_out, wav_stride, wav_lens);
    QTTS_API_END
}

int qtts_codec_stream_open(qtts_model* m, int B, int max_chunk_frames, qtts_codec_stream** out) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && out, ERR_BAD_ARG, "qtts_codec_stream_open: null argument");
    *out = reinterpret_cast<qtts_codec_stream*>(codec_stream_open(m->m, B, max_chunk_frames));
    QTTS_API_END
}

int qtts_codec_stream_push(qtts_codec_stream* s, const uint32_t* codes, int n_frames, float* wav_out, int64_t wav_stride,
                           int64_t* n_samples) {
    QTTS_API_BEGIN
    QTTS_CHECK(s && codes && wav_out && n_samples, ERR_BAD_ARG, "qtts_codec_stream_push: null argument");
    codec_stream_push(*reinterpret_cast<CodecStream*>(s), codes, n_frames, wav_out, wav_stride, n_samples);
    QTTS_API_END
}

void qtts_codec_stream_close(qtts_codec_stream* s) { codec_stream_close(reinterpret_cast<CodecStream*>(s)); }

int qtts_encode_audio(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, int num_quantizers,
                      uint32_t* codes_out, int T_cap, int* T_out) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && audio && n_samples && codes_out && T_out, ERR_BAD_ARG, "qtts_encode_audio: null argument");
    run_encode_audio(m->m, audio, n_samples, B, max_samples, num_quantizers, codes_out, T_cap, T_out);
    QTTS_API_END
}

int qtts_mel(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride,
             int* n_frames) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && audio && n_samples && mel_out && n_frames, ERR_BAD_ARG, "qtts_mel: null argument");
    run_mel(m->m, audio, n_samples, B, max_samples, mel_out, frames_stride, n_frames);
    QTTS_API_END
}

int qtts_mel_noalloc(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride,
                     int* n_frames) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && audio && n_samples && mel_out && n_frames, ERR_BAD_ARG, "qtts_mel_noalloc: null argument");
    run_mel_noalloc(m->m, audio, n_samples, B, max_samples, mel_out, frames_stride, n_frames);
    QTTS_API_END
}

int64_t qtts_mel_noalloc_capacity(void) { return (int64_t)mel_scratch_bytes(); }

int qtts_speaker_embed(qtts_model* m, const float* audio, const int* n_samples, int B, int max_samples, float* emb_out) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && audio && n_samples && emb_out, ERR_BAD_ARG, "qtts_speaker_embed: null argument");
    run_speaker_embed(m->m, audio, n_samples, B, max_samples, emb_out);
    QTTS_API_END
}

int qtts_speaker_embed_mel(qtts_model* m, const float* mel, const int* n_frames, int B, int frames_stride, float* emb_out) {
    QTTS_API_BEGIN
    QTTS_CHECK(m && mel && n_frames && emb_out, ERR_BAD_ARG, "qtts_speaker_embed_mel: null argument");
    run_speaker_embed_mel(m->m, mel, n_frames, B, frames_stride, emb_out);
    QTTS_API_END
}

// Diagnostics: average duration (microseconds, HIP events) of `iters` back-to-back launches of the weight-streaming
// GEMM on random-free zero... buffers of the given shape; used by the benchmark to report the kernel's roofline.
int qtts_bench_gemm(qtts_ctx* ctx, int dtype, int M, int N, int K, int KS, int iters, double* avg_us) {
    QTTS_API_BEGIN
    QTTS_CHECK(ctx && avg_us && iters >= 1, ERR_BAD_ARG, "qtts_bench_gemm: bad argument");
    QTTS_HIP(hipSetDevice(ctx->c.device));
    const size_t es = dtype_size(dtype);
    const int Mpad = gemm_pad_rows(M);
    if (KS <= 0) KS = gemm_pick_ks(M, N, K, dtype);
    // rotate over several weight copies so that consecutive launches do not hit the Infinity Cache
    const size_t wbytes = (size_t)N * K * es;
    const int copies = (int)std::max<size_t>(1, std::min<size_t>(16, ((size_t)600 << 20) / wbytes + 1));
    DevBuf W, X, out;
    W.alloc(wbytes * copies);
    X.alloc((size_t)Mpad * K * es);
    out.alloc((size_t)KS * Mpad * N * 4);
    QTTS_HIP(hipMemsetAsync(W.p, 0x3c, wbytes * copies, ctx->c.stream));
    QTTS_HIP(hipMemsetAsync(X.p, 0x3c, (size_t)Mpad * K * es, ctx->c.stream));
    ScopedEvent e0, e1;
    for (int i = 0; i < 3; ++i) launch_gemm(ctx->c.stream, dtype, (char*)W.p + wbytes * (i % copies), X.p, out.as<float>(), M, N, K, KS, K);
    QTTS_HIP(hipEventRecord(e0, ctx->c.stream));
    for (int i = 0; i < iters; ++i) launch_gemm(ctx->c.stream, dtype, (char*)W.p + wbytes * (i % copies), X.p, out.as<float>(), M, N, K, KS, K);
    QTTS_HIP(hipEventRecord(e1, ctx->c.stream));
    QTTS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    QTTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1000.0 / iters;
    QTTS_API_END
}

int qtts_bench_stream(qtts_ctx* ctx, long bytes, int blocks, int pattern, int unroll, int ld_bytes, int iters, double* avg_us) {
    QTTS_API_BEGIN
    QTTS_HIP(hipSetDevice(ctx->c.device));
    const int copies = (int)std::max<long>(1, std::min<long>(16, ((long)600 << 20) / bytes + 1));
    DevBuf src, sink;
    src.alloc((size_t)bytes * copies);
    sink.alloc(64);
    QTTS_HIP(hipMemsetAsync(src.p, 0x11, (size_t)bytes * copies, ctx->c.stream));
    ScopedEvent e0, e1;
    for (int i = 0; i < 3; ++i) launch_stream(ctx->c.stream, (char*)src.p + (size_t)bytes * (i % copies), sink.as<unsigned>(), bytes, blocks, pattern, unroll, ld_bytes);
    QTTS_HIP(hipEventRecord(e0, ctx->c.stream));
    for (int i = 0; i < iters; ++i) launch_stream(ctx->c.stream, (char*)src.p + (size_t)bytes * (i % copies), sink.as<unsigned>(), bytes, blocks, pattern, unroll, ld_bytes);
    QTTS_HIP(hipEventRecord(e1, ctx->c.stream));
    QTTS_HIP(hipEventSynchronize(e1));
    float ms = 0;
    QTTS_HIP(hipEventElapsedTime(&ms, e0, e1));
    *avg_us = (double)ms * 1000.0 / iters;
    QTTS_API_END
}

int qtts_model_info(qtts_model* m, int* hidden, int* vocab, int* groups, int* total_upsample, int* enc_dim) {
    QTTS_API_BEGIN
    QTTS_CHECK(m, ERR_BAD_ARG, "qtts_model_info: null model");
    if (hidden) *hidden = m->m.cfg.talker.hidden;
    if (vocab) *vocab = m->m.cfg.vocab;
    if (groups) *groups = m->m.cfg.groups;
    if (total_upsample) *total_upsample = m->m.codec.present ? m->m.codec.total_upsample() : 0;
    if (enc_dim) *enc_dim = m->m.cfg.has_speaker_encoder ? m->m.cfg.enc_dim : 0;
    QTTS_API_END
}

}  // extern "C"
