// Mel front end + ECAPA-TDNN speaker encoder on the GPU
// (qwen_tts/src/audio/mel.rs:81-158, qwen_tts/src/nn/speaker_encoder.rs:565-722).
//
// Every conv is a tap-addressed GEMM over channels-last strips (k_gemm.hip) followed by the shared fused epilogue
// (k_codec.hip: bias, ReLU / tanh, channel-slice writes); pooling, squeeze-excitation and the attentive statistics
// are small per-(utterance, channel) kernels (k_speaker.hip).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

#include "generate.hpp"
#include "prep.hpp"

namespace qtts {

namespace {
constexpr int NFFT = 1024, HOP = 256, NMEL = 128, NBIN = NFFT / 2 + 1, MELPAD = (NFFT - HOP) / 2;
constexpr long SLACK = CONV_HALO_SLACK + 64;

// create_mel_filterbank (audio/mel.rs:236-327): Slaney mel scale, f64 construction, f32 storage, area normalisation
std::vector<float> mel_filterbank_T(int sample_rate, double fmin, double fmax) {
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double hz) { return hz < min_log_hz ? hz / f_sp : min_log_mel + std::log(hz / min_log_hz) / logstep; };
    auto mel_to_hz = [&](double mel) { return mel < min_log_mel ? mel * f_sp : min_log_hz * std::exp((mel - min_log_mel) * logstep); };
    const double mmin = hz_to_mel(fmin), mmax = hz_to_mel(fmax);
    std::vector<double> hz((size_t)NMEL + 2);
    for (int i = 0; i <= NMEL + 1; ++i) hz[(size_t)i] = mel_to_hz(mmin + (mmax - mmin) * (double)i / (double)(NMEL + 1));
    std::vector<float> fbT((size_t)NBIN * NMEL, 0.f);  // transposed [bin][mel] so that a wave reads consecutive mels
    for (int m = 0; m < NMEL; ++m) {
        const double fl = hz[(size_t)m], fc = hz[(size_t)m + 1], fr = hz[(size_t)m + 2];
        const double ld = fc - fl, ud = fr - fc;
        const float enorm = 2.0f / (float)(fr - fl);
        for (int b = 0; b < NBIN; ++b) {
            const double freq = (double)b * (double)sample_rate / (double)NFFT;
            const double lower = ld > 0.0 ? (freq - fl) / ld : 0.0;
            const double upper = ud > 0.0 ? (fr - freq) / ud : 0.0;
            fbT[(size_t)b * NMEL + m] = (float)std::max(std::min(lower, upper), 0.0) * enorm;
        }
    }
    return fbT;
}
}  // namespace

struct SpeakerState {
    // mel constants
    DevBuf window, twiddle, fbT;
    // ECAPA weights
    bool have_encoder = false;
    GemmW block0;
    int pad0 = 0;
    struct Block {
        GemmW tdnn1, tdnn2;
        std::vector<GemmW> res;
        DevBuf se_w1, se_b1, se_w2, se_b2;
        int C = 0, CP = 0, cs = 0, pad = 0, se = 0;
    };
    std::vector<Block> blocks;
    GemmW mfa, asp_a, asp_conv;
    DevBuf asp_wb, asp_b, fc_w, fc_b;
    int Cm = 0, CmP = 0, Ca = 0, CaP = 0, enc_dim = 0, catP = 0, P = 0;
    // run-time buffers
    DevBuf audio, nsamp, lens, mel_f32, melT, h0, a, r, t2, tmp, cat, mm, att, acc, vec1, vec2, vec3, emb;
};

void finalize_speaker(Model& m) {
    QTTS_HIP(hipSetDevice(m.ctx->device));
    auto st = std::make_shared<SpeakerState>();
    SpeakerState& s = *st;
    {   // mel constants (mel.rs:54-61 periodic Hann in f32; twiddles from f64)
        std::vector<float> win((size_t)NFFT), tw((size_t)NFFT);
        for (int n = 0; n < NFFT; ++n) {
            const float x = 2.0f * 3.14159265358979323846f * (float)n / (float)NFFT;
            win[(size_t)n] = 0.5f * (1.0f - std::cos(x));
        }
        for (int k = 0; k < NFFT / 2; ++k) {
            const double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)NFFT;
            tw[(size_t)2 * k] = (float)std::cos(ang);
            tw[(size_t)2 * k + 1] = (float)std::sin(ang);
        }
        upload(s.window, win, F32);
        upload(s.twiddle, tw, F32);
        upload(s.fbT, mel_filterbank_T(24000, 0.0, 12000.0), F32);
    }
    if (m.cfg.has_speaker_encoder && m.has("speaker_encoder.blocks.0.conv.weight")) {
        const ModelCfg& c = m.cfg;
        const std::string P = "speaker_encoder.";
        const int nblk = (int)c.enc_channels.size() - 2;
        QTTS_CHECK(nblk >= 1 && c.enc_kernels.size() == c.enc_channels.size() && c.enc_dilations.size() == c.enc_channels.size(), ERR_SHAPE,
                   "speaker encoder: inconsistent channel / kernel / dilation lists");
        QTTS_CHECK(c.mel_dim == NMEL, ERR_SHAPE, "speaker encoder: mel_dim must be 128");
        conv_weight(m, P + "blocks.0.conv", c.enc_dilations[0], s.block0);
        QTTS_CHECK(s.block0.taps % 2 == 1, ERR_SHAPE, "speaker encoder: 'same' padding needs odd kernels");
        s.pad0 = (s.block0.taps - 1) * c.enc_dilations[0] / 2;
        s.P = std::max(1, s.pad0);
        int Cprev = c.enc_channels[0];
        s.catP = 0;
        for (int i = 1; i <= nblk; ++i) {
            const std::string B = P + "blocks." + std::to_string(i) + ".";
            s.blocks.emplace_back();
            SpeakerState::Block& b = s.blocks.back();
            b.C = c.enc_channels[(size_t)i];
            b.CP = pad64(b.C);
            QTTS_CHECK(b.C % c.res2net_scale == 0 && b.C % 64 == 0, ERR_SHAPE, "speaker encoder: block width must be a multiple of 64");
            b.cs = b.C / c.res2net_scale;
            QTTS_CHECK(b.cs <= 64 && b.cs % 4 == 0, ERR_SHAPE, "speaker encoder: Res2Net chunk wider than 64 channels");
            conv_weight(m, B + "tdnn1.conv", 1, b.tdnn1);
            conv_weight(m, B + "tdnn2.conv", 1, b.tdnn2);
            QTTS_CHECK(b.tdnn1.taps == 1 && b.tdnn2.taps == 1 && b.tdnn1.CinP == pad64(Cprev) && b.tdnn1.CoutP == b.CP && b.tdnn2.CinP == b.CP &&
                           b.tdnn2.CoutP == b.CP,
                       ERR_SHAPE, "speaker encoder: TDNN shapes do not chain");
            const int k = c.enc_kernels[(size_t)i], dil = c.enc_dilations[(size_t)i];
            QTTS_CHECK(k % 2 == 1, ERR_SHAPE, "speaker encoder: 'same' padding needs odd kernels");
            b.pad = (k - 1) * dil / 2;
            s.P = std::max(s.P, b.pad);
            for (int j = 0; j + 1 < c.res2net_scale; ++j) {
                b.res.emplace_back();
                conv_weight(m, B + "res2net_block.blocks." + std::to_string(j) + ".conv", dil, b.res.back());
                QTTS_CHECK(b.res.back().taps == k && b.res.back().CinP == 64 && b.res.back().CoutP == 64, ERR_SHAPE,
                           "speaker encoder: Res2Net conv has the wrong shape");
            }
            std::vector<float> w1 = host_f32(m, B + "se_block.conv1.weight"), w2 = host_f32(m, B + "se_block.conv2.weight");
            b.se = (int)m.get(B + "se_block.conv1.weight").shape[0];
            QTTS_CHECK((int)w1.size() == b.se * b.C && (int)w2.size() == b.C * b.se, ERR_SHAPE, "speaker encoder: SE block has the wrong shape");
            upload(b.se_w1, w1, F32);
            upload(b.se_w2, w2, F32);
            upload(b.se_b1, host_f32(m, B + "se_block.conv1.bias"), F32);
            upload(b.se_b2, host_f32(m, B + "se_block.conv2.bias"), F32);
            s.catP += b.CP;
            Cprev = b.C;
        }
        conv_weight(m, P + "mfa.conv", c.enc_dilations.back(), s.mfa);
        QTTS_CHECK(s.mfa.taps == 1 && s.mfa.CinP == s.catP, ERR_SHAPE, "speaker encoder: MFA conv must be 1x1 over the concatenated blocks");
        s.Cm = c.enc_channels.back();
        s.CmP = s.mfa.CoutP;
        QTTS_CHECK(s.Cm % 64 == 0, ERR_SHAPE, "speaker encoder: MFA width must be a multiple of 64");
        {   // asp.tdnn over cat(x, mean, std): the x part is a GEMM, the per-utterance (mean, std) part a bias vector
            const Tensor& t = m.get(P + "asp.tdnn.conv.weight");
            QTTS_CHECK(t.shape.size() == 3 && t.shape[1] == 3 * s.Cm && t.shape[2] == 1, ERR_SHAPE, "speaker encoder: asp.tdnn has the wrong shape");
            s.Ca = (int)t.shape[0];
            s.CaP = pad64(s.Ca);
            const std::vector<float> w = host_f32(m, P + "asp.tdnn.conv.weight");
            std::vector<float> wa((size_t)s.CaP * s.CmP, 0.f), wb((size_t)s.Ca * 2 * s.Cm);
            for (int o = 0; o < s.Ca; ++o) {
                for (int i = 0; i < s.Cm; ++i) wa[(size_t)o * s.CmP + i] = w[(size_t)o * 3 * s.Cm + i];
                for (int i = 0; i < 2 * s.Cm; ++i) wb[(size_t)o * 2 * s.Cm + i] = w[(size_t)o * 3 * s.Cm + s.Cm + i];
            }
            upload(s.asp_a.w, wa, m.dtype);
            s.asp_a.N = s.CaP; s.asp_a.K = s.CmP; s.asp_a.CinP = s.CmP; s.asp_a.CoutP = s.CaP;
            upload(s.asp_wb, wb, F32);
            upload(s.asp_b, host_f32(m, P + "asp.tdnn.conv.bias"), F32);
        }
        conv_weight(m, P + "asp.conv", 1, s.asp_conv);
        QTTS_CHECK(s.asp_conv.taps == 1 && s.asp_conv.CinP == s.CaP && s.asp_conv.CoutP == s.CmP, ERR_SHAPE, "speaker encoder: asp.conv has the wrong shape");
        {
            const Tensor& t = m.get(P + "fc.weight");
            QTTS_CHECK(t.shape.size() == 3 && t.shape[1] == 2 * s.Cm && t.shape[2] == 1, ERR_SHAPE, "speaker encoder: fc has the wrong shape");
            s.enc_dim = (int)t.shape[0];
            upload(s.fc_w, host_f32(m, P + "fc.weight"), F32);
            upload(s.fc_b, host_f32(m, P + "fc.bias"), F32);
        }
        s.have_encoder = true;
    }
    m.spk_state = st;
}

namespace {

struct Batch {
    int B = 0, max_frames = 0;
    std::vector<int> frames;
};

// uploads the audio and computes frame counts; frames = (n + 2*384 - 1024) / 256 + 1 (mel.rs:100-103)
Batch stage_audio(Model& m, SpeakerState& s, const float* audio, const int* n_samples, int B, int max_samples) {
    QTTS_CHECK(B >= 1 && max_samples >= 1, ERR_BAD_ARG, "mel: bad batch shape");
    Batch b;
    b.B = B;
    b.frames.resize((size_t)B);
    for (int u = 0; u < B; ++u) {
        QTTS_CHECK(n_samples[u] > MELPAD && n_samples[u] <= max_samples, ERR_BAD_ARG,
                   "mel: every clip needs more than 384 samples (reflect padding) and at most max_samples");
        b.frames[(size_t)u] = (n_samples[u] + 2 * MELPAD - NFFT) / HOP + 1;
        QTTS_CHECK(b.frames[(size_t)u] >= 1, ERR_BAD_ARG, "mel: audio shorter than one STFT frame");
        b.max_frames = std::max(b.max_frames, b.frames[(size_t)u]);
    }
    s.audio.ensure((size_t)B * max_samples * 4);
    s.nsamp.ensure((size_t)B * 4);
    s.lens.ensure((size_t)B * 4);
    copy_sync(s.audio.p, audio, (size_t)B * max_samples * 4, hipMemcpyHostToDevice, m.ctx->stream);
    copy_sync(s.nsamp.p, n_samples, (size_t)B * 4, hipMemcpyHostToDevice, m.ctx->stream);
    copy_sync(s.lens.p, b.frames.data(), (size_t)B * 4, hipMemcpyHostToDevice, m.ctx->stream);
    (void)m;
    return b;
}

}  // namespace

void run_mel(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride, int* n_frames) {
    std::lock_guard<std::mutex> lock(m.locks->speaker);  // sessions of one model share these workspaces
    QTTS_CHECK(m.finalized && m.spk_state, ERR_STATE, "model is not finalized");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    SpeakerState& s = *m.spk_state;
    hipStream_t st = m.ctx->stream;
    const Batch b = stage_audio(m, s, audio, n_samples, B, max_samples);
    QTTS_CHECK(frames_stride >= b.max_frames, ERR_SHAPE, "mel: frames_stride too small");
    s.mel_f32.ensure((size_t)B * frames_stride * NMEL * 4);
    QTTS_HIP(hipMemsetAsync(s.mel_f32.p, 0, (size_t)B * frames_stride * NMEL * 4, st));
    launch_mel(st, F32, s.audio.as<float>(), s.nsamp.as<int>(), B, max_samples, b.max_frames, s.window.as<float>(), s.twiddle.as<float>(),
               s.fbT.as<float>(), s.mel_f32.as<float>(), frames_stride, nullptr, 0, 0);
    QTTS_HIP(hipMemcpyAsync(mel_out, s.mel_f32.p, (size_t)B * frames_stride * NMEL * 4, hipMemcpyDeviceToHost, st));
    QTTS_HIP(hipStreamSynchronize(st));
    for (int u = 0; u < B; ++u) n_frames[u] = b.frames[(size_t)u];
}

// ------------------------------------------------------------------------------------------------
// run_mel for callers that must not allocate: the staging copies of the clips, of their lengths and of the result live in
// one fixed device array per GPU instead of buffers grown on demand.  Every call on a GPU shares that GPU's array: calls from
// any model or session on that GPU take turns (a per-GPU lock held until the result is on the host); a call overwrites what it
// reads back, so nothing has to be cleared between calls except the result region, which the call itself zeroes.
constexpr size_t MEL_SCRATCH_BYTES = (size_t)16 << 20;
static __device__ __align__(256) unsigned char scratch[MEL_SCRATCH_BYTES];
namespace {
constexpr int MAX_DEVICES = 64;
unsigned char* g_scratch_ptr[MAX_DEVICES] = {};  // device address of `scratch` on each GPU (the array's name is only a host-side handle)
std::mutex g_scratch_mu;
std::mutex g_scratch_use[MAX_DEVICES];           // held for a whole call: every model and session on a GPU shares that GPU's array
size_t pad256(size_t n) { return (n + 255) / 256 * 256; }
}  // namespace

// Looks up this GPU's copy of the array once; called with `device` current (qtts_init, and again by the entry point in case
// the context was created by other means).
unsigned char* mel_scratch(int device) {
    QTTS_CHECK(device >= 0 && device < MAX_DEVICES, ERR_BAD_ARG, "mel scratch: device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        QTTS_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        QTTS_CHECK(p != nullptr, ERR_HIP, "mel scratch: no device address for the scratch array");
        g_scratch_ptr[device] = (unsigned char*)p;
    }
    return g_scratch_ptr[device];
}

size_t mel_scratch_bytes() { return MEL_SCRATCH_BYTES; }

void run_mel_noalloc(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride, int* n_frames) {
    QTTS_CHECK(m.finalized && m.spk_state, ERR_STATE, "model is not finalized");
    QTTS_CHECK(m.ctx->device >= 0 && m.ctx->device < MAX_DEVICES, ERR_BAD_ARG, "mel scratch: device ordinal out of range");
    std::lock_guard<std::mutex> use(g_scratch_use[m.ctx->device]);  // the array belongs to the GPU, not to the model
    QTTS_CHECK(B >= 1 && max_samples >= 1 && frames_stride >= 1, ERR_BAD_ARG, "mel: bad batch shape");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    SpeakerState& s = *m.spk_state;
    hipStream_t st = m.ctx->stream;
    int max_frames = 0;
    for (int u = 0; u < B; ++u) {
        QTTS_CHECK(n_samples[u] > MELPAD && n_samples[u] <= max_samples, ERR_BAD_ARG,
                   "mel: every clip needs more than 384 samples (reflect padding) and at most max_samples");
        n_frames[u] = (n_samples[u] + 2 * MELPAD - NFFT) / HOP + 1;
        QTTS_CHECK(n_frames[u] >= 1, ERR_BAD_ARG, "mel: audio shorter than one STFT frame");
        max_frames = std::max(max_frames, n_frames[u]);
    }
    QTTS_CHECK(frames_stride >= max_frames, ERR_SHAPE, "mel: frames_stride too small");
    const size_t b_len = pad256((size_t)B * 4), b_audio = pad256((size_t)B * max_samples * 4), b_mel = (size_t)B * frames_stride * NMEL * 4;
    QTTS_CHECK(b_len + b_audio + b_mel <= MEL_SCRATCH_BYTES, ERR_SHAPE,
               "mel (no-allocation entry point): clips + result need " + std::to_string(b_len + b_audio + b_mel) +
                   " bytes of staging, the fixed scratch array holds " + std::to_string(MEL_SCRATCH_BYTES) + "; use qtts_mel or a smaller batch");
    unsigned char* base = mel_scratch(m.ctx->device);
    int* d_len = (int*)base;
    float* d_audio = (float*)(base + b_len);
    float* d_mel = (float*)(base + b_len + b_audio);
    QTTS_HIP(hipMemcpyAsync(d_len, n_samples, (size_t)B * 4, hipMemcpyHostToDevice, st));
    QTTS_HIP(hipMemcpyAsync(d_audio, audio, (size_t)B * max_samples * 4, hipMemcpyHostToDevice, st));
    QTTS_HIP(hipMemsetAsync(d_mel, 0, b_mel, st));
    launch_mel(st, F32, d_audio, d_len, B, max_samples, max_frames, s.window.as<float>(), s.twiddle.as<float>(), s.fbT.as<float>(), d_mel,
               frames_stride, nullptr, 0, 0);
    QTTS_HIP(hipMemcpyAsync(mel_out, d_mel, b_mel, hipMemcpyDeviceToHost, st));
    QTTS_HIP(hipStreamSynchronize(st));
}

namespace {
void speaker_embed_impl(Model& m, const float* audio, const int* n_samples, int max_samples, const float* mel_in, const int* n_frames,
                        int frames_stride, int B, float* emb_out);
}

void run_speaker_embed(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* emb_out) {
    std::lock_guard<std::mutex> lock(m.locks->speaker);  // sessions of one model share these workspaces
    speaker_embed_impl(m, audio, n_samples, max_samples, nullptr, nullptr, 0, B, emb_out);
}

// ConditionalGeneration::encode_speaker on a mel the caller already holds (Model::create_voice_clone_prompt_from_mel,
// model/mod.rs:579-612): mel_in [B][frames_stride][128] f32 (log-mel, frames x mels), n_frames[b] valid frames.
void run_speaker_embed_mel(Model& m, const float* mel_in, const int* n_frames, int B, int frames_stride, float* emb_out) {
    std::lock_guard<std::mutex> lock(m.locks->speaker);  // sessions of one model share these workspaces
    QTTS_CHECK(B >= 1 && frames_stride >= 1, ERR_BAD_ARG, "speaker encoder: bad mel batch shape");
    for (int u = 0; u < B; ++u) QTTS_CHECK(n_frames[u] >= 1 && n_frames[u] <= frames_stride, ERR_BAD_ARG, "speaker encoder: n_frames out of range");
    speaker_embed_impl(m, nullptr, nullptr, 0, mel_in, n_frames, frames_stride, B, emb_out);
}

namespace {
void speaker_embed_impl(Model& m, const float* audio, const int* n_samples, int max_samples, const float* mel_in, const int* n_frames,
                        int frames_stride, int B, float* emb_out) {
    QTTS_CHECK(m.finalized && m.spk_state, ERR_STATE, "model is not finalized");
    SpeakerState& s = *m.spk_state;
    QTTS_CHECK(s.have_encoder, ERR_STATE,
               "Speaker encoder not available - this model doesn't support voice cloning. Use a Base model for voice cloning.");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    hipStream_t st = m.ctx->stream;
    const int dt = m.dtype;
    const size_t es = dtype_size(dt);
    Batch b;
    if (mel_in) {
        b.B = B;
        b.frames.assign(n_frames, n_frames + B);
        for (int f : b.frames) b.max_frames = std::max(b.max_frames, f);
        s.lens.ensure((size_t)B * 4);
        copy_sync(s.lens.p, b.frames.data(), (size_t)B * 4, hipMemcpyHostToDevice, m.ctx->stream);
    } else {
        b = stage_audio(m, s, audio, n_samples, B, max_samples);
    }
    const int Tm = b.max_frames, P = s.P, U = B;
    for (int u = 0; u < U; ++u)
        QTTS_CHECK(b.frames[(size_t)u] > s.P, ERR_BAD_ARG, "speaker encoder: clip too short for the reflect padding of its convolutions");
    const long S = Tm + 2L * P, M = U * S;
    const int* d_len = s.lens.as<int>();
    int Cmax = std::max({NMEL, s.block0.CoutP, s.catP, s.CmP, s.CaP});
    for (auto& blk : s.blocks) Cmax = std::max(Cmax, blk.CP);
    const size_t rows = (size_t)(M + SLACK);
    s.melT.ensure(rows * NMEL * es);
    s.h0.ensure(rows * s.block0.CoutP * es);
    int CPmax = 0;
    for (auto& blk : s.blocks) CPmax = std::max(CPmax, blk.CP);
    s.a.ensure(rows * CPmax * es);
    s.r.ensure(rows * CPmax * es);
    s.t2.ensure(rows * CPmax * es);
    s.tmp.ensure(rows * 64 * es);
    s.cat.ensure(rows * s.catP * es);
    s.mm.ensure(rows * s.CmP * es);
    s.att.ensure(rows * s.CaP * es);
    s.acc.ensure((size_t)((M + 63) / 64 * 64) * Cmax * 4);
    s.vec1.ensure((size_t)U * 2 * std::max(s.CmP, CPmax) * 4);
    s.vec2.ensure((size_t)U * std::max({s.CmP, CPmax, s.CaP}) * 4);
    s.vec3.ensure((size_t)U * std::max({s.CmP, CPmax, s.CaP}) * 4);
    s.emb.ensure((size_t)U * s.enc_dim * 4);
    float* acc = s.acc.as<float>();
    auto zero = [&](DevBuf& d, int C) { QTTS_HIP(hipMemsetAsync(d.p, 0, rows * (size_t)C * es, st)); };
    // conv / linear as one launch: tap-addressed GEMM with bias + activation applied in registers
    auto conv = [&](const GemmW& g, const void* X, int ldx, int act, void* out, int ld_out, const float* bias_u = nullptr, int Cvalid = 0) {
        ConvEpi e{};
        e.M = M; e.N = g.N; e.r = 1; e.Cout = g.CoutP; e.bias = g.bias.p ? g.bias.as<float>() : nullptr; e.bias_u = bias_u;
        e.Cvalid = Cvalid; e.act = act; e.lens = d_len; e.S_in = S; e.S_out = S; e.P_out = P; e.ld_out = ld_out; e.out = out;
        launch_gemm_rows_epi(st, dt, g.wm(), X, g.K, ldx, g.taps, (long)g.dil * ldx, e);
    };
    auto at = [&](const DevBuf& d, long row, int ld) { return (const char*)d.p + (size_t)row * ld * es; };

    // mel -> strips (model dtype, like the reference's cast before the encoder) with reflected borders
    if (mel_in) {
        // the caller's mel, cast to the model dtype and laid out as strips with P leading rows
        std::vector<float> strips(rows * NMEL, 0.f);
        for (int u = 0; u < U; ++u)
            std::memcpy(&strips[((size_t)u * S + P) * NMEL], mel_in + (size_t)u * frames_stride * NMEL, (size_t)b.frames[(size_t)u] * NMEL * 4);
        QTTS_HIP(hipStreamSynchronize(st));
        upload(s.melT, strips, dt);
    } else {
        zero(s.melT, NMEL);
        launch_mel(st, dt, s.audio.as<float>(), s.nsamp.as<int>(), U, max_samples, Tm, s.window.as<float>(), s.twiddle.as<float>(),
                   s.fbT.as<float>(), nullptr, 0, s.melT.p, S, P);
    }
    if (s.pad0 > 0) launch_reflect_fill(st, dt, s.melT.p, d_len, U, S, P, s.pad0, NMEL);
    // blocks.0: TDNN k = 5 + ReLU
    zero(s.h0, s.block0.CoutP);
    conv(s.block0, at(s.melT, P - s.pad0, NMEL), NMEL, 2, s.h0.p, s.block0.CoutP);
    // SE-Res2Net blocks; block i writes its output into channel slice i of `cat`
    zero(s.cat, s.catP);
    const void* hin = s.h0.p;
    int ld_in = s.block0.CoutP, Cin = m.cfg.enc_channels[0];
    int off = 0;
    for (size_t bi = 0; bi < s.blocks.size(); ++bi) {
        const SpeakerState::Block& blk = s.blocks[bi];
        // tdnn1 (1x1 + ReLU)
        zero(s.a, blk.CP);
        conv(blk.tdnn1, (const char*)hin + (size_t)P * ld_in * es, ld_in, 2, s.a.p, blk.CP);
        // Res2Net: chunk 0 passes through, chunk j = conv(chunk_j [+ out_{j-1}]) + ReLU
        zero(s.r, blk.CP);
        launch_slice_copy(st, dt, s.a.p, d_len, U, Tm, S, P, blk.CP, blk.cs, s.r.p);
        for (size_t j = 1; j <= blk.res.size(); ++j) {
            const GemmW& g = blk.res[j - 1];
            launch_res2net_prep(st, dt, s.a.p, j >= 2 ? s.r.p : nullptr, d_len, U, Tm, S, P, blk.pad, blk.CP, (int)j * blk.cs, blk.cs, s.tmp.p);
            conv(g, at(s.tmp, P - blk.pad, 64), 64, 2, (char*)s.r.p + (size_t)j * blk.cs * es, blk.CP, nullptr, blk.cs);
        }
        // tdnn2 (1x1 + ReLU)
        zero(s.t2, blk.CP);
        conv(blk.tdnn2, at(s.r, P, blk.CP), blk.CP, 2, s.t2.p, blk.CP);
        // squeeze-excitation: mean over time -> 1x1 + ReLU -> 1x1 + sigmoid -> scale, then the block residual
        launch_time_stats(st, dt, s.t2.p, blk.CP, nullptr, 0, nullptr, d_len, U, S, P, 0, blk.C, 0, s.vec1.as<float>());
        launch_small_fc(st, blk.se_w1.as<float>(), blk.C, blk.se_b1.as<float>(), s.vec1.as<float>(), blk.C, U, blk.se, blk.C, 2, s.vec2.as<float>(), blk.se);
        launch_small_fc(st, blk.se_w2.as<float>(), blk.se, blk.se_b2.as<float>(), s.vec2.as<float>(), blk.se, U, blk.C, blk.se, 4, s.vec3.as<float>(), blk.C);
        launch_se_apply(st, dt, s.t2.p, blk.CP, s.vec3.as<float>(), Cin == blk.C ? hin : nullptr, ld_in, d_len, U, Tm, S, P, blk.C,
                        (char*)s.cat.p + (size_t)off * es, s.catP);
        hin = (const char*)s.cat.p + (size_t)off * es;
        ld_in = s.catP;
        Cin = blk.C;
        off += blk.CP;
    }
    // MFA: 1x1 TDNN over the concatenated block outputs
    zero(s.mm, s.CmP);
    conv(s.mfa, at(s.cat, P, s.catP), s.catP, 2, s.mm.p, s.CmP);
    // attentive statistics pooling (lengths = None: every step of the clip counts)
    launch_time_stats(st, dt, s.mm.p, s.CmP, nullptr, 0, nullptr, d_len, U, S, P, 0, s.Cm, 1, s.vec1.as<float>());
    launch_small_fc(st, s.asp_wb.as<float>(), 2 * s.Cm, s.asp_b.as<float>(), s.vec1.as<float>(), 2 * s.Cm, U, s.Ca, 2 * s.Cm, 0, s.vec2.as<float>(),
                    s.CaP);
    zero(s.att, s.CaP);
    s.asp_a.taps = 1;
    s.asp_a.dil = 1;
    conv(s.asp_a, at(s.mm, P, s.CmP), s.CmP, 3, s.att.p, s.CaP, s.vec2.as<float>(), (s.Ca + 3) / 4 * 4);
    launch_gemm_rows(st, dt, s.asp_conv.wm(), at(s.att, P, s.CaP), acc, M, s.asp_conv.N, s.asp_conv.K, s.CaP, 1, 0);
    // softmax over time of (asp.conv output + bias), read straight from the GEMM output (its rows carry no leading padding)
    launch_time_stats(st, dt, s.mm.p, s.CmP, acc, s.asp_conv.N, s.asp_conv.bias.as<float>(), d_len, U, S, P, 0, s.Cm, 2, s.vec1.as<float>());
    launch_small_fc(st, s.fc_w.as<float>(), 2 * s.Cm, s.fc_b.as<float>(), s.vec1.as<float>(), 2 * s.Cm, U, s.enc_dim, 2 * s.Cm, 0, s.emb.as<float>(),
                    s.enc_dim);
    QTTS_HIP(hipMemcpyAsync(emb_out, s.emb.p, (size_t)U * s.enc_dim * 4, hipMemcpyDeviceToHost, st));
    QTTS_HIP(hipStreamSynchronize(st));
}
}  // namespace

}  // namespace qtts
