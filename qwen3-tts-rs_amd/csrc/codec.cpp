// 12.5 Hz codec decoder on the GPU: codes -> 24 kHz waveform (qwen_tts/src/audio/decoder/v2.rs:163-286).
//
// Activations are channels-last strips (one per utterance or per decode chunk) so that every conv /
// transposed conv / linear of the decoder is one large-M GEMM (k_gemm.hip, taps = row offsets) followed by one
// fused epilogue (k_codec.hip).  Weights are rearranged once at load time into [N][taps * Cin] rows with the
// channel counts padded to multiples of 64; padded channels carry zeros end to end.
#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstring>

#include "generate.hpp"
#include "prep.hpp"
#include "runner.hpp"

namespace qtts {

namespace {

constexpr int PAGE = 64;

}  // namespace

struct CodecState {
    int Q = 0, csize = 0, DcP = 0, latP = 0;
    DevBuf table;  // [Q][csize][DcP] f32: codebook rows with the group's output_proj applied
    GemmW pre_conv, in_proj, out_proj, dec0;
    StackW xf;
    struct Up {
        GemmW ct, pw1, pw2;
        DevBuf dw_w, dw_b, ln_w, ln_b, gamma;
    };
    std::vector<Up> ups;
    struct Unit {
        Snake a1, a2;
        GemmW c1, c2;
        DevBuf w1n, w2n;  // narrow blocks: the two convs in the order conv_unit96_kernel reads them
    };
    struct Block {
        Snake s0;
        GemmW ct;
        Unit u[3];
        // narrow: the block's streams keep their real 96 channels per row instead of the 128 the padded layout gives them (bf16 runs,
        // last block only: nothing downstream but the output conv reads its rows)
        bool narrow = false;
    };
    std::vector<Block> blocks;
    Snake sfinal;
    DevBuf final_w;  // [7][CP] f32
    DevBuf final_wn; // [7][96] f32: the same weights for a narrow last block
    float final_b = 0.f;
    int final_CP = 0;
    int maxpad = 0;  // leading zero rows of the strips inside the vocoder blocks
};

// Run-time buffers of the decoder (grown on demand, kept between calls).  Every model handle - the model itself and each of its
// sessions - owns a set, so that the codec decodes of several sessions run side by side instead of queueing for one workspace
// (a session that had to wait for another one's decode left its stream idle for up to a whole decode).  A handle is driven by
// one host thread at a time, so no lock is needed.  xf is a view of the transformer's weights with RoPE tables of its own.
struct CodecScratch {
    Workspace ws;
    StackW xf;
    DevBuf kv, toks, tiles, meta, codes, A, Bf, R, Rf, Rf2, acc, wav;
    // the buffers of the last streaming session that was closed on this handle, for the next one to take over: a hipFree waits
    // for every queue of the GPU to drain (15-60 ms with other sessions decoding), a session per utterance batch would pay it
    // every time
    DevBuf spare_kv, spare_latent;
};

static CodecScratch& codec_scratch(Model& m) {
    if (!m.codec_scratch) {
        auto w = std::make_shared<CodecScratch>();
        const StackW& b = m.codec_state->xf;
        w->xf.cfg = b.cfg;
        w->xf.layers = b.layers;
        w->xf.final_norm = b.final_norm;
        w->xf.max_pos = 0;  // RoPE tables grow on demand, per handle
        m.codec_scratch = w;
    }
    return *m.codec_scratch;
}

// Streaming decode session: the transformer's keys/values live in a ring per row, and the last `ctx_cap` frames of codes and
// of transformer output are kept so that the convolutional stack of the next push sees real history instead of zero padding.
struct CodecStream {
    Model* m = nullptr;
    int B = 0, max_chunk = 0, ctx_cap = 0, ring_pages = 0;
    long frames = 0;                  // frames pushed so far (every row advances together)
    int C = 0;                        // context frames held (min(frames, ctx_cap))
    std::vector<uint32_t> ctx_codes;  // [B][C][Q]
    DevBuf latent_ctx;                // [B][ctx_cap][latP] model dtype
    DevBuf kv;                        // [layers][(B + 1) * ring_pages] pages; row B takes the writes of rows that are not tokens
};

void finalize_codec(Model& m) {
    if (!m.codec.present || !m.has("decoder.pre_conv.conv.weight")) return;
    QTTS_HIP(hipSetDevice(m.ctx->device));
    auto st = std::make_shared<CodecState>();
    CodecState& s = *st;
    const CodecCfg& c = m.codec;
    QTTS_CHECK(c.hidden % 64 == 0 && c.inter % 64 == 0 && (c.heads * c.head_dim) % 64 == 0, ERR_SHAPE,
               "codec transformer sizes must be multiples of 64");
    QTTS_CHECK(c.codebook_dim % 2 == 0 && c.num_quantizers >= 1, ERR_SHAPE, "codec: bad quantizer shape");
    s.Q = c.num_quantizers;
    s.csize = c.codebook_size;
    s.DcP = pad64(c.codebook_dim);
    s.latP = pad64(c.latent);

    // --- RVQ tables (quantizer.rs:56-59, 417-434): (embedding_sum / max(usage, 1e-5)) then the group's 1x1 output_proj
    {
        const int half = c.codebook_dim / 2, Dc = c.codebook_dim;
        std::vector<float> table((size_t)s.Q * s.csize * s.DcP, 0.f);
        const std::vector<float> p_first = host_f32(m, "decoder.quantizer.rvq_first.output_proj.weight");
        std::vector<float> p_rest;
        if (s.Q > 1) p_rest = host_f32(m, "decoder.quantizer.rvq_rest.output_proj.weight");
        QTTS_CHECK((int)p_first.size() == Dc * half, ERR_SHAPE, "rvq output_proj has the wrong shape");
        std::vector<float> e((size_t)half);
        for (int q = 0; q < s.Q; ++q) {
            const std::string p = q == 0 ? "decoder.quantizer.rvq_first.vq.layers.0._codebook."
                                         : "decoder.quantizer.rvq_rest.vq.layers." + std::to_string(q - 1) + "._codebook.";
            const std::vector<float> usage = host_f32(m, p + "cluster_usage"), sum = host_f32(m, p + "embedding_sum");
            QTTS_CHECK((int)usage.size() == s.csize && (int)sum.size() == s.csize * half, ERR_SHAPE, "codebook has the wrong shape");
            const std::vector<float>& proj = q == 0 ? p_first : p_rest;
            for (int r = 0; r < s.csize; ++r) {
                const float u = std::max(usage[(size_t)r], 1e-5f);
                for (int i = 0; i < half; ++i) e[(size_t)i] = sum[(size_t)r * half + i] / u;
                float* dst = &table[((size_t)q * s.csize + r) * s.DcP];
                for (int o = 0; o < Dc; ++o) {
                    float acc = 0.f;
                    for (int i = 0; i < half; ++i) acc += proj[(size_t)o * half + i] * e[(size_t)i];
                    dst[o] = acc;
                }
            }
        }
        upload(s.table, table, F32);
    }
    conv_weight(m, "decoder.pre_conv.conv", 1, s.pre_conv);
    linear_weight(m, "decoder.pre_transformer.input_proj", s.in_proj);
    linear_weight(m, "decoder.pre_transformer.output_proj", s.out_proj);
    QTTS_CHECK(s.pre_conv.CinP == s.DcP && s.pre_conv.CoutP == s.latP && s.in_proj.CinP == s.latP && s.in_proj.CoutP == c.hidden &&
                   s.out_proj.CinP == c.hidden && s.out_proj.CoutP == s.latP,
               ERR_SHAPE, "codec front-end shapes do not chain");
    // --- sliding-window transformer (audio/tokenizer/v2/transformer.rs)
    StackCfg sc;
    sc.hidden = c.hidden; sc.inter = c.inter; sc.layers = c.layers; sc.nh = c.heads; sc.nkv = c.kv_heads; sc.d = c.head_dim;
    sc.eps = c.eps; sc.theta = c.rope_theta; sc.window = c.window; sc.bias = c.attention_bias;
    QTTS_CHECK(sc.nh % sc.nkv == 0 && (sc.d == 64 || sc.d == 128), ERR_SHAPE, "codec attention geometry is not supported");
    build_stack(m, s.xf, sc, "decoder.pre_transformer.", false, true);
    // --- upsample stages: transposed conv (kernel == stride) + ConvNeXt block
    for (size_t i = 0; i < c.upsampling_ratios.size(); ++i) {
        const std::string p = "decoder.upsample." + std::to_string(i) + ".";
        s.ups.emplace_back();
        CodecState::Up& u = s.ups.back();
        convT_weight(m, p + "0.conv", c.upsampling_ratios[i], u.ct);
        QTTS_CHECK(u.ct.taps == 1 && u.ct.CinP == s.latP && u.ct.CoutP == s.latP, ERR_SHAPE, "upsample conv has the wrong shape");
        const Tensor& dw = m.get(p + "1.dwconv.conv.weight");
        QTTS_CHECK(dw.shape.size() == 3 && dw.shape[0] == c.latent && dw.shape[1] == 1 && dw.shape[2] == 7, ERR_SHAPE,
                   "ConvNeXt depthwise conv must be (C, 1, 7)");
        QTTS_CHECK(c.latent == s.latP, ERR_SHAPE, "ConvNeXt LayerNorm needs latent_dim to be a multiple of 64");
        padded_vec(m, p + "1.dwconv.conv.weight", s.latP * 7, 0.f, u.dw_w);
        padded_vec(m, p + "1.dwconv.conv.bias", s.latP, 0.f, u.dw_b);
        padded_vec(m, p + "1.norm.weight", s.latP, 0.f, u.ln_w);
        padded_vec(m, p + "1.norm.bias", s.latP, 0.f, u.ln_b);
        padded_vec(m, p + "1.gamma", s.latP, 0.f, u.gamma);
        linear_weight(m, p + "1.pwconv1", u.pw1);
        linear_weight(m, p + "1.pwconv2", u.pw2);
        QTTS_CHECK(u.pw1.CinP == s.latP && u.pw2.CoutP == s.latP && u.pw1.CoutP == u.pw2.CinP, ERR_SHAPE, "ConvNeXt MLP has the wrong shape");
    }
    // --- vocoder
    conv_weight(m, "decoder.decoder.0.conv", 1, s.dec0);
    QTTS_CHECK(s.dec0.taps == 7 && s.dec0.CinP == s.latP, ERR_SHAPE, "decoder.0 must be a k=7 conv over the latent");
    int CP = s.dec0.CoutP;
    s.maxpad = 0;
    const int dils[3] = {1, 3, 9};
    for (size_t b = 0; b < c.upsample_rates.size(); ++b) {
        const std::string p = "decoder.decoder." + std::to_string(b + 1) + ".block.";
        s.blocks.emplace_back();
        CodecState::Block& blk = s.blocks.back();
        load_snake(m, p + "0.", CP, blk.s0);
        convT_weight(m, p + "1.conv", c.upsample_rates[b], blk.ct);
        QTTS_CHECK(blk.ct.taps == 2 && blk.ct.CinP == CP, ERR_SHAPE, "decoder block conv must have kernel == 2 * stride");
        CP = blk.ct.CoutP;
        for (int u = 0; u < 3; ++u) {
            const std::string q = p + std::to_string(u + 2) + ".";
            load_snake(m, q + "act1.", CP, blk.u[u].a1);
            load_snake(m, q + "act2.", CP, blk.u[u].a2);
            conv_weight(m, q + "conv1.conv", dils[u], blk.u[u].c1);
            conv_weight(m, q + "conv2.conv", 1, blk.u[u].c2);
            QTTS_CHECK(blk.u[u].c1.taps == 7 && blk.u[u].c2.taps == 1 && blk.u[u].c1.CinP == CP && blk.u[u].c1.CoutP == CP &&
                           blk.u[u].c2.CinP == CP && blk.u[u].c2.CoutP == CP,
                       ERR_SHAPE, "residual unit has the wrong shape");
            s.maxpad = std::max(s.maxpad, 6 * dils[u]);
        }
        // QTTS_UNIT96=0: the last block in the padded layout like the others
        static const bool narrow_off = getenv("QTTS_UNIT96") && atoi(getenv("QTTS_UNIT96")) == 0;
        const int Creal = (int)m.get(p + "1.conv.weight").shape[1];
        if (!narrow_off && m.dtype == BF16 && b + 1 == c.upsample_rates.size() && Creal == 96 && CP == 128) {
            blk.narrow = true;
            for (int u = 0; u < 3; ++u) {
                const std::string q = p + std::to_string(u + 2) + ".";
                const std::vector<float> w1 = host_f32(m, q + "conv1.conv.weight"), w2 = host_f32(m, q + "conv2.conv.weight");
                QTTS_CHECK(w1.size() == (size_t)96 * 96 * 7 && w2.size() == (size_t)96 * 96, ERR_SHAPE, "residual unit has the wrong shape");
                std::vector<uint16_t> p1((size_t)6 * 21 * 64 * 8), p2((size_t)6 * 3 * 64 * 8);
                conv_unit96_pack(w1.data(), 7, p1.data(), [](float v) { return f32_to_bf16(v); });
                conv_unit96_pack(w2.data(), 1, p2.data(), [](float v) { return f32_to_bf16(v); });
                blk.u[u].w1n.alloc(p1.size() * 2);
                blk.u[u].w2n.alloc(p2.size() * 2);
                QTTS_HIP(hipMemcpy(blk.u[u].w1n.p, p1.data(), p1.size() * 2, hipMemcpyHostToDevice));
                QTTS_HIP(hipMemcpy(blk.u[u].w2n.p, p2.data(), p2.size() * 2, hipMemcpyHostToDevice));
            }
        }
    }
    const int nb = (int)c.upsample_rates.size();
    load_snake(m, "decoder.decoder." + std::to_string(nb + 1) + ".", CP, s.sfinal);
    {
        const std::string key = "decoder.decoder." + std::to_string(nb + 2) + ".conv.";
        const Tensor& t = m.get(key + "weight");
        QTTS_CHECK(t.shape.size() == 3 && t.shape[0] == 1 && t.shape[2] == 7 && pad64((int)t.shape[1]) == CP, ERR_SHAPE,
                   "final conv must be (1, C, 7)");
        const int C = (int)t.shape[1];
        const std::vector<float> w = host_f32(m, key + "weight");
        std::vector<float> out((size_t)7 * CP, 0.f);
        for (int ci = 0; ci < C; ++ci)
            for (int k = 0; k < 7; ++k) out[(size_t)k * CP + ci] = w[(size_t)ci * 7 + k];
        upload(s.final_w, out, F32);
        if (!s.blocks.empty() && s.blocks.back().narrow) {
            std::vector<float> outn((size_t)7 * 96, 0.f);
            for (int ci = 0; ci < C; ++ci)
                for (int k = 0; k < 7; ++k) outn[(size_t)k * 96 + ci] = w[(size_t)ci * 7 + k];
            upload(s.final_wn, outn, F32);
        }
        s.final_b = host_f32(m, key + "bias")[0];
        s.final_CP = CP;
    }
    m.codec_state = st;
}

namespace {

struct Piece {  // one strip: a whole utterance or one window of the chunked decode
    int row, start, len, drop_frames;
    int64_t out_off;  // where its samples land in the row's output
};

// Strip geometry of one group of pieces: stage 0 = frame rate, then one stage per upsample block, then one per
// vocoder block.  S = strip length in rows, P = leading zero rows, Tmax = longest valid length.
struct Geometry {
    int U = 0, nup = 0, nb = 0, nstage = 0;
    std::vector<long> Tmax, S;
    std::vector<int> P;
    size_t act_elems = 0, acc_elems = 0, res_elems = 0;
    size_t lo_elems = 0;  // largest tensor that carries a remainder plane in bf16 runs (the low-rate stages)
    long wav_max = 0;
    size_t bytes(size_t es) const {
        return (act_elems + lo_elems) * es * 3 + (acc_elems + res_elems * (size_t)(f32_units ? 2 : 1)) * 4 + (size_t)U * (size_t)wav_max * 4;
    }
    bool f32_units = false;  // the fused residual units read the f32 stream (two f32 buffers, written in turn)
};

constexpr long SLACK_ROWS = CONV_HALO_SLACK + 64;  // readable rows past the last strip: 64-row GEMM tile padding + tap offsets
constexpr int SPLIT_BLOCKS = 2;  // vocoder blocks whose transposed conv reads an input with a remainder plane (bf16 runs)

// bf16 runs: the fused residual units make their activated input from the f32 stream, so that no activated copy travels between
// the layers of a vocoder block (a quarter fewer bytes written per row in its narrow stages, paid for with a second read of the f32
// rows: measured 3.5 % of a decode; the values are the same bit for bit).  QTTS_UNIT_F32IN=0: activated copies as before.
bool unit_f32in(int dtype) {
    static const int mode = getenv("QTTS_UNIT_F32IN") ? atoi(getenv("QTTS_UNIT_F32IN")) : 1;
    return dtype == BF16 && mode != 0;
}

Geometry make_geometry(const CodecCfg& c, const CodecState& s, int U, int Tm, int dtype) {
    Geometry g;
    g.f32_units = unit_f32in(dtype);
    g.U = U;
    g.nup = (int)c.upsampling_ratios.size();
    g.nb = (int)c.upsample_rates.size();
    g.nstage = 1 + g.nup + g.nb;
    g.Tmax.resize((size_t)g.nstage);
    g.S.resize((size_t)g.nstage);
    g.P.resize((size_t)g.nstage);
    g.Tmax[0] = Tm;
    g.P[0] = 2;  // pre_conv k = 3
    for (int i = 0; i < g.nup; ++i) {
        g.Tmax[(size_t)i + 1] = g.Tmax[(size_t)i] * c.upsampling_ratios[(size_t)i];
        g.P[(size_t)i + 1] = 6;  // ConvNeXt depthwise k = 7, decoder.0 k = 7
    }
    for (int b = 0; b < g.nb; ++b) {
        const int i = g.nup + b;
        g.Tmax[(size_t)i + 1] = std::max<long>((g.Tmax[(size_t)i] - 1) * c.upsample_rates[(size_t)b], 1);
        g.P[(size_t)i + 1] = s.maxpad;
    }
    for (int i = 0; i < g.nstage; ++i) g.S[(size_t)i] = g.P[(size_t)i] + g.Tmax[(size_t)i];
    auto need = [&](long rows, int width, bool split = false) {
        g.act_elems = std::max(g.act_elems, (size_t)(rows + SLACK_ROWS) * (size_t)width);
        if (split) g.lo_elems = std::max(g.lo_elems, (size_t)(rows + SLACK_ROWS) * (size_t)width);
    };
    auto need_acc = [&](long M, int N) { g.acc_elems = std::max(g.acc_elems, (size_t)((M + 63) / 64 * 64) * (size_t)N); };
    const long M0 = U * g.S[0];
    need(M0, std::max({s.DcP, s.latP}), true);
    need_acc(M0, s.latP);
    for (int i = 0; i < g.nup; ++i) {
        const long Mout = U * g.S[(size_t)i + 1];
        need(Mout, s.ups[(size_t)i].pw1.CoutP);
        need(Mout, s.latP, true);
    }
    need(U * g.S[(size_t)g.nup], s.dec0.CoutP, true);
    for (int b = 0; b < g.nb; ++b) {
        const long Mout = U * g.S[(size_t)g.nup + b + 1];
        need(Mout, s.blocks[(size_t)b].ct.CoutP, b + 1 < SPLIT_BLOCKS);
        g.res_elems = std::max(g.res_elems, (size_t)(Mout + SLACK_ROWS) * (size_t)s.blocks[(size_t)b].ct.CoutP);
    }
    g.wav_max = g.Tmax[(size_t)g.nstage - 1];
    return g;
}

struct Decoder {
    Model& m;
    CodecState& s;     // weights
    CodecScratch& w;   // this handle's buffers
    hipStream_t st;
    int dt;
    size_t es;

    explicit Decoder(Model& model)
        : m(model), s(*model.codec_state), w(codec_scratch(model)), st(model.ctx->stream), dt(model.dtype), es(dtype_size(model.dtype)) {}

    // A conv is issued in two steps so that the call sites read like the reference's forward(): gemm() names the
    // weights and the input view, epilogue() launches the GEMM with that element-wise tail applied in registers.
    struct Pending {
        const GemmW* g = nullptr;
        const void* X = nullptr;
        long M = 0;
        int ldx = 0;
        long x_lo = 0;  // remainder plane of X (bf16 runs, the low-rate stages)
    } pend;
    void gemm(const GemmW& g, const void* X, long M, int ldx, long x_lo = 0) { pend = Pending{&g, X, M, ldx, x_lo}; }
    // QTTS_CODEC_PROFILE=1: HIP events around every conv launch, one table per group on stderr (measurement aid)
    struct Prof {
        hipEvent_t a, b;
        long M;
        int N, K, taps, dil;
    };
    std::vector<Prof> prof;
    bool profiling = getenv("QTTS_CODEC_PROFILE") != nullptr;
    void epilogue(ConvEpi e) {
        QTTS_CHECK(pend.g != nullptr && e.M == pend.M && e.N == pend.g->N, ERR_STATE, "codec decode: epilogue does not match the pending GEMM");
        e.acc = nullptr;
        Prof pr{};
        if (profiling) {
            pr = Prof{nullptr, nullptr, pend.M, pend.g->N, pend.g->K, pend.g->taps, pend.g->dil};
            QTTS_HIP(hipEventCreate(&pr.a));
            QTTS_HIP(hipEventCreate(&pr.b));
            QTTS_HIP(hipEventRecord(pr.a, st));
        }
        launch_gemm_rows_epi(st, dt, pend.g->wm(), pend.X, pend.g->K, pend.ldx, pend.g->taps, (long)pend.g->dil * pend.ldx, e, pend.x_lo);
        if (profiling) {
            QTTS_HIP(hipEventRecord(pr.b, st));
            prof.push_back(pr);
        }
        pend = Pending{};
    }
    void report() {
        for (Prof& p : prof) {
            float ms = 0;
            (void)hipEventElapsedTime(&ms, p.a, p.b);
            const double fl = 2.0 * (double)p.M * p.N * p.K;
            fprintf(stderr, "codec conv M=%9ld N=%5d K=%5d taps=%2d dil=%d  %8.3f ms  %7.1f TFLOP/s  in+out %6.2f GB/s-equiv\n", p.M, p.N, p.K, p.taps,
                    p.dil, ms, fl / ms / 1e9, ((double)p.M * (p.K / p.taps + p.N) * es) / ms / 1e6);
            (void)hipEventDestroy(p.a);
            (void)hipEventDestroy(p.b);
        }
        prof.clear();
    }
    void run_group(const uint32_t* d_codes, int T_max, const std::vector<Piece>& ps, float* wav_out, int64_t wav_stride, CodecStream* sc = nullptr);
};

void Decoder::run_group(const uint32_t* d_codes, int T_max, const std::vector<Piece>& ps, float* wav_out, int64_t wav_stride, CodecStream* sc) {
    const CodecCfg& c = m.codec;
    const int U = (int)ps.size();
    int Tm = 0;
    for (auto& p : ps) Tm = std::max(Tm, p.len);
    const Geometry g = make_geometry(c, s, U, Tm, dt);
    const int nup = g.nup, nb = g.nb, nstage = g.nstage;
    const std::vector<long>& S = g.S;
    const std::vector<int>& P = g.P;
    // ---- valid lengths per stage
    std::vector<std::vector<int>> L((size_t)nstage, std::vector<int>((size_t)U));
    for (int u = 0; u < U; ++u) L[0][(size_t)u] = ps[(size_t)u].len;
    for (int i = 0; i < nup; ++i)
        for (int u = 0; u < U; ++u) L[(size_t)i + 1][(size_t)u] = L[(size_t)i][(size_t)u] * c.upsampling_ratios[(size_t)i];
    for (int b = 0; b < nb; ++b)
        for (int u = 0; u < U; ++u) {
            const int i = nup + b;
            L[(size_t)i + 1][(size_t)u] = (L[(size_t)i][(size_t)u] - 1) * c.upsample_rates[(size_t)b];
            QTTS_CHECK(L[(size_t)i + 1][(size_t)u] >= 1, ERR_SHAPE, "codec decode: sequence too short for the vocoder's transposed convs");
        }
    // ---- device metadata: rows, starts, per-stage valid lengths and (valid - 1) for the two-tap transposed convs
    std::vector<int> meta;
    auto push = [&](const std::vector<int>& v) {
        const size_t off = meta.size();
        meta.insert(meta.end(), v.begin(), v.end());
        return off;
    };
    std::vector<int> rows((size_t)U), starts((size_t)U);
    for (int u = 0; u < U; ++u) {
        rows[(size_t)u] = ps[(size_t)u].row;
        starts[(size_t)u] = ps[(size_t)u].start;
    }
    const size_t o_rows = push(rows), o_starts = push(starts);
    std::vector<size_t> o_len((size_t)nstage), o_lenm1((size_t)nstage);
    for (int i = 0; i < nstage; ++i) {
        o_len[(size_t)i] = push(L[(size_t)i]);
        std::vector<int> t = L[(size_t)i];
        for (auto& v : t) v -= 1;
        o_lenm1[(size_t)i] = push(t);
    }
    w.meta.ensure(meta.size() * 4);
    QTTS_HIP(hipMemcpyAsync(w.meta.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, st));
    QTTS_HIP(hipStreamSynchronize(st));  // `meta` is a local
    const int* d_meta = w.meta.as<int>();
    auto d_len = [&](int i) { return d_meta + o_len[(size_t)i]; };
    auto d_lenm1 = [&](int i) { return d_meta + o_lenm1[(size_t)i]; };

    // Only the leading rows of every strip (the causal padding the next conv reads) must be zero: rows past a strip's valid
    // length are never read by a valid row, and whatever a tile computes from them is masked in the epilogue.
    // bf16 runs keep a remainder plane (see launch_gemm) behind every activation buffer and use it for the tensors of the
    // low-rate stages - everything up to the second vocoder block's transposed conv - where bf16 rounding of the activations
    // is what limits the waveform SNR (each of them alone costs 42-46 dB; the high-rate vocoder tensors 49 dB and more).
    // Their convs run through the tap-addressed GEMM with both planes; these stages are a few percent of the decoder's FLOPs.
    const long lo = split_activations(dt) ? (long)g.act_elems : 0;
    auto zero_pad_rows = [&](void* buf, long S_rows, int P_rows, int C, bool both = false) {
        if (P_rows > 0) {
            QTTS_HIP(hipMemset2DAsync(buf, (size_t)S_rows * C * es, 0, (size_t)P_rows * C * es, (size_t)U, st));
            if (both && lo)
                QTTS_HIP(hipMemset2DAsync((char*)buf + (size_t)lo * es, (size_t)S_rows * C * es, 0, (size_t)P_rows * C * es, (size_t)U, st));
        }
    };
    w.A.ensure((g.act_elems + (lo ? g.lo_elems : 0)) * es);
    w.Bf.ensure((g.act_elems + (lo ? g.lo_elems : 0)) * es);
    w.R.ensure((g.act_elems + (lo ? g.lo_elems : 0)) * es);
    w.Rf.ensure(g.res_elems * 4);  // the vocoder's residual stream is kept in f32
    if (g.f32_units) w.Rf2.ensure(g.res_elems * 4);
    w.acc.ensure(g.acc_elems * 4);
    w.wav.ensure((size_t)U * (size_t)g.wav_max * 4);

    ScopedEvent ev0, ev1;
    QTTS_HIP(hipEventRecord(ev0, st));
    // ================= stage 0: RVQ lookup -> pre_conv -> transformer =================
    const long M0 = U * S[0];
    const int Hd = c.hidden;
    QTTS_HIP(hipMemsetAsync(w.A.p, 0, (size_t)(M0 + SLACK_ROWS) * s.DcP * es, st));
    if (lo) QTTS_HIP(hipMemsetAsync((char*)w.A.p + (size_t)lo * es, 0, (size_t)(M0 + SLACK_ROWS) * s.DcP * es, st));
    launch_rvq_gather(st, dt, s.table.as<float>(), d_codes, d_meta + o_rows, d_meta + o_starts, d_len(0), U, Tm, s.Q, s.csize, s.DcP,
                      T_max, S[0], P[0], w.A.p, s.DcP, lo);
    // pre_conv (k = 3): acc row (u, t) reads strip rows t .. t + 2 (two leading zero rows)
    gemm(s.pre_conv, w.A.p, M0, s.DcP, lo);
    QTTS_HIP(hipMemsetAsync(w.Bf.p, 0, (size_t)(M0 + SLACK_ROWS) * s.latP * es, st));
    if (lo) QTTS_HIP(hipMemsetAsync((char*)w.Bf.p + (size_t)lo * es, 0, (size_t)(M0 + SLACK_ROWS) * s.latP * es, st));
    {
        ConvEpi e{};
        e.M = M0; e.N = s.pre_conv.N; e.r = 1; e.Cout = s.pre_conv.CoutP; e.bias = s.pre_conv.bias.as<float>();
        e.lens = d_len(0); e.S_in = S[0]; e.S_out = S[0]; e.P_out = 0; e.ld_out = s.latP; e.out = w.Bf.p; e.out_lo = lo;
        epilogue(e);
    }
    // transformer over every strip row (rows past a strip's length are computed and ignored)
    {
        QTTS_CHECK(M0 < (1L << 30), ERR_SHAPE, "codec decode: group too large");
        const int M = (int)M0;
        const size_t rp = (size_t)gemm_pad_rows(M) + 64;
        const int nqkv = (c.heads + 2 * c.kv_heads) * c.head_dim, nq = c.heads * c.head_dim;
        const int Nmax = std::max({nqkv, 2 * c.inter, Hd, s.latP});
        w.ws.x.ensure(rp * Hd * 4);
        w.ws.alloc_inputs(rp, (size_t)Hd, (size_t)nq, (size_t)c.inter, es, split_activations(dt));
        w.ws.part.ensure((size_t)Nmax * std::max(rp, (size_t)GEMM_MAX_KS * std::min<size_t>(rp, 512)) * 4);
        w.ws.q.ensure(rp * nq * 4);
        // pages: one private range per strip + one shared dump page for the padding rows
        const int pp = sc ? sc->ring_pages : (Tm + PAGE - 1) / PAGE;
        // a streaming push holds C context frames in front of the new ones: only the new frames are tokens, at the
        // positions they have in the whole sequence; their keys/values join the session's ring
        const int Cs = sc ? sc->C : 0;
        const long pos0 = sc ? sc->frames - Cs : 0;
        std::vector<Tok> toks((size_t)M);
        for (int u = 0; u < U; ++u)
            for (long r = 0; r < S[0]; ++r) {
                Tok& t = toks[(size_t)(u * S[0] + r)];
                if (r >= Cs && r < L[0][(size_t)u]) {
                    const int pos = (int)(pos0 + r), lo = std::max(0, pos + 1 - c.window);
                    t = Tok{u, pos, lo, pos + 1 - lo};
                } else {
                    const int slot = (int)((u * S[0] + r) % PAGE);
                    t = Tok{U, slot, slot, 1};
                }
            }
        const size_t page_elems = (size_t)2 * c.kv_heads * PAGE * c.head_dim;
        if (!sc) w.kv.ensure(((size_t)U + 1) * pp * page_elems * es);  // row U: dump pages for the padding rows
        w.toks.ensure(toks.size() * sizeof(Tok));
        QTTS_HIP(hipMemcpyAsync(w.toks.p, toks.data(), toks.size() * sizeof(Tok), hipMemcpyHostToDevice, st));
        // the strips' rows are runs of consecutive positions: banded attention in tiles of up to 64 queries on the matrix cores
        std::vector<AttnTile> tiles;
        build_attn_tiles(toks.data(), M, tiles);
        w.tiles.ensure(tiles.size() * sizeof(AttnTile));
        QTTS_HIP(hipMemcpyAsync(w.tiles.p, tiles.data(), tiles.size() * sizeof(AttnTile), hipMemcpyHostToDevice, st));
        QTTS_HIP(hipStreamSynchronize(st));  // toks and tiles are locals
        // one-shot decodes reuse one layer's worth of pages for every layer; a session keeps all layers
        KVView kv{w.kv.p, 0, PAGE, c.kv_heads, c.head_dim, pp, nullptr};
        if (sc) {
            kv.base = sc->kv.p;
            kv.layer_stride = ((size_t)U + 1) * pp * page_elems;
            kv.page_mask = pp - 1;
        }
        const int pos_end = (int)(pos0 + Tm);
        if (w.xf.max_pos < std::max(pos_end, PAGE)) {
            int n = 256;
            while (n < pos_end) n *= 2;
            std::vector<float> cs, sn;
            rope_table_simple(c.head_dim, c.rope_theta, n, cs, sn);
            w.xf.cos.alloc(cs.size() * 4);
            w.xf.sin.alloc(sn.size() * 4);
            copy_sync(w.xf.cos.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, m.ctx->stream);
            copy_sync(w.xf.sin.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, m.ctx->stream);
            w.xf.max_pos = n;
        }
        Runner run(m, w.ws, nullptr);
        run.tiles = w.tiles.as<AttnTile>();
        run.n_tiles = (int)tiles.size();
        // input_proj (+bias) starts the residual stream
        int ks = run.ks_for(M, Hd, s.latP);
        run.gemm(s.in_proj.wm(), w.Bf.p, M, Hd, s.latP, ks, s.latP, lo);
        launch_resid_norm(st, dt, nullptr, w.ws.part.as<float>(), ks, run.kstride(M, Hd), s.in_proj.bias.as<float>(), nullptr,
                          w.xf.layers[0].ln1, c.eps, w.ws.x.as<float>(), w.ws.xn.p, nullptr, M, Hd, w.ws.lo_xn);
        run.stack(w.xf, M, w.toks.as<Tok>(), kv, w.ws.xn.p, nullptr, sc != nullptr, false, w.ws.lo_xn);
        // output_proj (+bias) -> strips without padding (the next op is a kernel == stride transposed conv)
        ks = run.ks_for(M, s.latP, Hd);
        run.gemm(s.out_proj.wm(), w.ws.xn.p, M, s.latP, Hd, ks, Hd, w.ws.lo_xn);
        const float* acc = w.ws.part.as<float>();
        if (ks != 1) {  // fold the split-K slabs so that the conv epilogue sees one [M][N] matrix
            launch_bias_act(st, w.ws.part.as<float>(), ks, run.kstride(M, s.latP), nullptr, 0, w.acc.p, F32, M, s.latP);
            acc = w.acc.as<float>();
        }
        ConvEpi e{};
        e.acc = acc; e.M = M0; e.N = s.latP; e.r = 1; e.Cout = s.latP; e.bias = s.out_proj.bias.as<float>();
        e.lens = d_len(0); e.S_in = S[0]; e.S_out = S[0]; e.P_out = 0; e.ld_out = s.latP; e.out = w.A.p; e.out_lo = lo;
        launch_conv_epilogue(st, dt, e);
        if (sc) {
            // rows [0, C) of every strip: the transformer output of the previous pushes; then keep this push's tail
            const size_t rowb = (size_t)s.latP * es, strip = (size_t)S[0] * rowb, slot = (size_t)sc->ctx_cap * rowb;
            const int keep = std::min(Tm, sc->ctx_cap);
            // the session's copy holds both planes of a split tensor, the second one sc->B * slot bytes behind the first
            for (int pl = 0; pl < (lo ? 2 : 1); ++pl) {
                char* a = (char*)w.A.p + (size_t)pl * lo * es;
                char* ctx = (char*)sc->latent_ctx.p + (size_t)pl * sc->B * slot;
                if (Cs > 0) QTTS_HIP(hipMemcpy2DAsync(a, strip, ctx, slot, (size_t)Cs * rowb, (size_t)U, hipMemcpyDeviceToDevice, st));
                QTTS_HIP(hipMemcpy2DAsync(ctx, slot, a + (size_t)(Tm - keep) * rowb, strip, (size_t)keep * rowb, (size_t)U, hipMemcpyDeviceToDevice, st));
            }
        }
    }
    // ================= upsample stages: the current tensor lives in A with Pcur leading rows per strip =================
    int Pcur = 0;
    for (int i = 0; i < nup; ++i) {
        const CodecState::Up& up = s.ups[(size_t)i];
        const long Min = U * S[(size_t)i], Mout = U * S[(size_t)i + 1];
        const int Pn = P[(size_t)i + 1];
        // transposed conv (kernel == stride): A -> R (strips with six leading zero rows for the depthwise conv)
        gemm(up.ct, (const char*)w.A.p + (size_t)Pcur * s.latP * es, Min, s.latP, lo);
        zero_pad_rows(w.R.p, S[(size_t)i + 1], Pn, s.latP, true);
        {
            ConvEpi e{};
            e.M = Min; e.N = up.ct.N; e.r = up.ct.r; e.Cout = up.ct.CoutP; e.bias = up.ct.bias.as<float>();
            e.lens = d_len(i); e.S_in = S[(size_t)i]; e.S_out = S[(size_t)i + 1]; e.P_out = Pn; e.ld_out = s.latP; e.out = w.R.p; e.out_lo = lo;
            epilogue(e);
        }
        // ConvNeXt: depthwise conv + LayerNorm -> A ; pwconv1 + GELU -> Bf ; pwconv2 * gamma + residual -> R (in place)
        launch_dwconv_ln(st, dt, w.R.p, up.dw_w.as<float>(), up.dw_b.as<float>(), up.ln_w.as<float>(), up.ln_b.as<float>(), d_len(i + 1), U,
                         (int)g.Tmax[(size_t)i + 1], S[(size_t)i + 1], Pn, s.latP, w.A.p, lo);
        gemm(up.pw1, (const char*)w.A.p + (size_t)Pn * s.latP * es, Mout, s.latP);
        {
            ConvEpi e{};
            e.M = Mout; e.N = up.pw1.N; e.r = 1; e.Cout = up.pw1.CoutP; e.bias = up.pw1.bias.as<float>();
            e.act = 1; e.lens = d_len(i + 1); e.S_in = S[(size_t)i + 1]; e.S_out = S[(size_t)i + 1]; e.P_out = 0; e.ld_out = up.pw1.CoutP;
            e.out = w.Bf.p;
            epilogue(e);
        }
        gemm(up.pw2, w.Bf.p, Mout, up.pw1.CoutP);
        {
            ConvEpi e{};
            e.M = Mout; e.N = up.pw2.N; e.r = 1; e.Cout = up.pw2.CoutP; e.bias = up.pw2.bias.as<float>();
            e.scale = up.gamma.as<float>(); e.resid = w.R.p; e.lens = d_len(i + 1); e.S_in = S[(size_t)i + 1]; e.S_out = S[(size_t)i + 1];
            e.P_out = Pn; e.ld_out = s.latP; e.out = w.R.p; e.resid_lo = lo; e.out_lo = lo;
            epilogue(e);
        }
        std::swap(w.A, w.R);  // the stage's result becomes the current tensor
        Pcur = Pn;
    }
    // ================= vocoder =================
    QTTS_CHECK(Pcur >= 6, ERR_STATE, "codec decode: decoder.0 needs six leading zero rows (no upsample stage configured?)");
    int CP = s.dec0.CoutP;
    {
        // decoder.0 (k = 7): A -> snake(block 1 .0) in Bf; the raw result is not needed again
        const long M = U * S[(size_t)nup];
        gemm(s.dec0, (const char*)w.A.p + (size_t)(Pcur - 6) * s.latP * es, M, s.latP, lo);
        ConvEpi e{};
        e.M = M; e.N = s.dec0.N; e.r = 1; e.Cout = CP; e.bias = s.dec0.bias.as<float>();
        e.lens = d_len(nup); e.S_in = S[(size_t)nup]; e.S_out = S[(size_t)nup]; e.P_out = 0; e.ld_out = CP;
        e.out2 = w.Bf.p; e.alpha_exp = s.blocks[0].s0.a.as<float>(); e.beta_den = s.blocks[0].s0.den.as<float>();
        e.out2_lo = lo;
        epilogue(e);
    }
    // Inside a block R is the residual stream and A the activated input of the next conv, both with maxpad leading
    // zero rows per strip.  `xin` is the activated input of the block's transposed conv: Bf for the first block,
    // afterwards the previous block's A viewed from its first valid row.
    const char* xin = (const char*)w.Bf.p;
    bool last_narrow = false, final_done = false;
    // QTTS_FINAL_FUSED=0: the output conv as a kernel of its own after a narrow last block, too
    static const bool final_fused = !(getenv("QTTS_FINAL_FUSED") && atoi(getenv("QTTS_FINAL_FUSED")) == 0);
    for (int b = 0; b < nb; ++b) {
        const CodecState::Block& blk = s.blocks[(size_t)b];
        const int si = nup + b, so = si + 1;
        const long Min = U * S[(size_t)si], Mout = U * S[(size_t)so];
        const int Pn = P[(size_t)so];
        const int CPo = blk.ct.CoutP;
        gemm(blk.ct, xin, Min, CP, b < SPLIT_BLOCKS ? lo : 0);
        // the block's units read the f32 stream where the fused kernel covers the block's width
        bool f32in = g.f32_units;
        for (int u = 0; u < 3; ++u) f32in = f32in && conv_unit_supported(dt, CPo, blk.u[u].c1.taps, blk.u[u].c1.dil);
        // a narrow block keeps 96 channels per row in its streams: the transposed conv computes its padded columns and stores the real ones
        const bool nar = blk.narrow && f32in && b + 1 == nb;
        const int CW = nar ? 96 : CPo;
        last_narrow = nar;
        zero_pad_rows(w.A.p, S[(size_t)so], Pn, CW);
        {
            ConvEpi e{};
            e.M = Min; e.N = blk.ct.N; e.r = blk.ct.r; e.Cout = CPo; e.bias = blk.ct.bias.as<float>();
            e.lens = d_lenm1(si); e.S_in = S[(size_t)si]; e.S_out = S[(size_t)so]; e.P_out = Pn; e.ld_out = CW;
            if (nar) e.Cvalid = CW;
            e.out_f32 = w.Rf.as<float>(); e.out2 = w.A.p; e.alpha_exp = blk.u[0].a1.a.as<float>(); e.beta_den = blk.u[0].a1.den.as<float>();
            if (f32in) e.out2 = nullptr;  // the first unit activates its own input from Rf
            epilogue(e);
        }
        // Residual units.  Where the fused kernel covers the block's width, each unit is one launch that reads the activated
        // tensor from one buffer and writes the next one to the other (a block of rows reads its neighbours' rows as halo, so
        // the unit cannot work in place); otherwise conv1 and conv2 are two launches through Bf.
        if (f32in) {
            // leading rows of every strip are the causal padding of the f32 streams the units read (SnakeBeta of 0 is 0)
            QTTS_HIP(hipMemset2DAsync(w.Rf.p, (size_t)S[(size_t)so] * CW * 4, 0, (size_t)Pn * CW * 4, (size_t)U, st));
            QTTS_HIP(hipMemset2DAsync(w.Rf2.p, (size_t)S[(size_t)so] * CW * 4, 0, (size_t)Pn * CW * 4, (size_t)U, st));
            float* rin = w.Rf.as<float>();
            float* rout = w.Rf2.as<float>();
            // QTTS_UNIT96_PAIR=0: every unit of a narrow block a launch of its own
            static const bool pair_on = !(getenv("QTTS_UNIT96_PAIR") && atoi(getenv("QTTS_UNIT96_PAIR")) == 0);
            int u_first = 0;
            if (nar && pair_on && blk.u[0].c1.dil == 1 && blk.u[1].c1.dil == 3) {
                // the first two units of a narrow block in one launch: the stream between them stays on the CU
                const CodecState::Unit &ua = blk.u[0], &ub = blk.u[1];
                ConvEpi e{};
                e.M = Mout; e.N = CW; e.r = 1; e.Cout = CW; e.bias = ua.c2.bias.as<float>();
                e.resid_f32 = rin; e.lens = d_len(so); e.S_in = S[(size_t)so]; e.S_out = S[(size_t)so]; e.P_out = Pn; e.ld_out = CW;
                e.out_f32 = rout;
                Prof pr{};
                if (profiling) {  // (reported as one conv of 16 "taps": two units)
                    pr = Prof{nullptr, nullptr, Mout, CPo, ua.c1.K + ua.c2.K + ub.c1.K + ub.c2.K, 2 * (ua.c1.taps + 1), 1};
                    QTTS_HIP(hipEventCreate(&pr.a));
                    QTTS_HIP(hipEventCreate(&pr.b));
                    QTTS_HIP(hipEventRecord(pr.a, st));
                }
                if (launch_conv_unit96_pair(st, ua.w1n.p, ua.w2n.p, 1, 3, ua.c1.bias.as<float>(), ua.a2.a.as<float>(), ua.a2.den.as<float>(), e,
                                            rin + (size_t)(Pn - 6) * CW, ua.a1.a.as<float>(), ua.a1.den.as<float>(), ub.w1n.p, ub.w2n.p,
                                            ub.c1.bias.as<float>(), ub.a2.a.as<float>(), ub.a2.den.as<float>(), ub.c2.bias.as<float>(),
                                            ub.a1.a.as<float>(), ub.a1.den.as<float>())) {
                    std::swap(rin, rout);
                    u_first = 2;
                    if (profiling) {
                        QTTS_HIP(hipEventRecord(pr.b, st));
                        prof.push_back(pr);
                    }
                }
            }
            for (int u = u_first; u < 3; ++u) {
                const CodecState::Unit& un = blk.u[u];
                ConvEpi e{};
                e.M = Mout; e.N = CW; e.r = 1; e.Cout = CW; e.bias = un.c2.bias.as<float>();
                e.resid_f32 = rin; e.lens = d_len(so); e.S_in = S[(size_t)so]; e.S_out = S[(size_t)so]; e.P_out = Pn; e.ld_out = CW;
                if (u < 2) {
                    e.out_f32 = rout;  // the next unit reads (and activates) this stream
                } else {           // after the last unit only the activated copy is read again
                    const Snake& nx = b + 1 < nb ? s.blocks[(size_t)b + 1].s0 : s.sfinal;
                    e.alpha_exp = nx.a.as<float>(); e.beta_den = nx.den.as<float>();
                    e.out2 = w.A.p;
                    e.out2_lo = b + 1 < SPLIT_BLOCKS ? lo : 0;
                }
                // the very last unit of a narrow last block takes the output conv along: its activated copy never leaves the CU
                Unit96Final fin{};
                const bool with_final = nar && u == 2 && b + 1 == nb && final_fused;
                if (with_final) {
                    fin.w = s.final_wn.as<float>(); fin.bias = s.final_b; fin.wav = w.wav.as<float>(); fin.wav_stride = g.wav_max;
                    e.out2 = nullptr;
                    e.out2_lo = 0;
                    final_done = true;
                }
                Prof pr{};
                if (profiling) {
                    pr = Prof{nullptr, nullptr, Mout, CPo, un.c1.K + un.c2.K, un.c1.taps + 1, un.c1.dil};
                    QTTS_HIP(hipEventCreate(&pr.a));
                    QTTS_HIP(hipEventCreate(&pr.b));
                    QTTS_HIP(hipEventRecord(pr.a, st));
                }
                const bool ok = nar ? launch_conv_unit96(st, un.w1n.p, un.w2n.p, un.c1.dil, un.c1.bias.as<float>(), un.a2.a.as<float>(), un.a2.den.as<float>(), e,
                                                         rin + (size_t)(Pn - 6 * un.c1.dil) * CW, un.a1.a.as<float>(), un.a1.den.as<float>(), with_final ? &fin : nullptr)
                                    : launch_conv_unit(st, dt, un.c1.wm(), un.c2.wm(), nullptr, CPo, un.c1.taps, un.c1.dil, un.c1.bias.as<float>(),
                                                       un.a2.a.as<float>(), un.a2.den.as<float>(), e, rin + (size_t)(Pn - 6 * un.c1.dil) * CPo,
                                                       un.a1.a.as<float>(), un.a1.den.as<float>());
                QTTS_CHECK(ok, ERR_STATE, "codec decode: fused residual unit refused a shape it reported as supported");
                if (profiling) {
                    QTTS_HIP(hipEventRecord(pr.b, st));
                    prof.push_back(pr);
                }
                std::swap(rin, rout);
            }
        }
        bool fuse = !f32in;
        for (int u = 0; u < 3; ++u) fuse = fuse && conv_unit_supported(dt, CPo, blk.u[u].c1.taps, blk.u[u].c1.dil);
        if (fuse) {
            zero_pad_rows(w.Bf.p, S[(size_t)so], Pn, CPo);
            DevBuf* in = &w.A;
            DevBuf* outb = &w.Bf;
            for (int u = 0; u < 3; ++u) {
                const CodecState::Unit& un = blk.u[u];
                ConvEpi e{};
                e.M = Mout; e.N = CPo; e.r = 1; e.Cout = CPo; e.bias = un.c2.bias.as<float>();
                e.resid_f32 = w.Rf.as<float>(); e.lens = d_len(so); e.S_in = S[(size_t)so]; e.S_out = S[(size_t)so]; e.P_out = Pn; e.ld_out = CPo;
                const Snake& nx = u < 2 ? blk.u[u + 1].a1 : (b + 1 < nb ? s.blocks[(size_t)b + 1].s0 : s.sfinal);
                e.alpha_exp = nx.a.as<float>(); e.beta_den = nx.den.as<float>();
                e.out_f32 = u < 2 ? w.Rf.as<float>() : nullptr;
                e.out2 = outb->p;
                e.out2_lo = (u == 2 && b + 1 < SPLIT_BLOCKS) ? lo : 0;
                Prof pr{};
                if (profiling) {
                    pr = Prof{nullptr, nullptr, Mout, CPo, un.c1.K + un.c2.K, un.c1.taps + 1, un.c1.dil};
                    QTTS_HIP(hipEventCreate(&pr.a));
                    QTTS_HIP(hipEventCreate(&pr.b));
                    QTTS_HIP(hipEventRecord(pr.a, st));
                }
                const bool ok = launch_conv_unit(st, dt, un.c1.wm(), un.c2.wm(), (const char*)in->p + (size_t)(Pn - 6 * un.c1.dil) * CPo * es, CPo,
                                                 un.c1.taps, un.c1.dil, un.c1.bias.as<float>(), un.a2.a.as<float>(), un.a2.den.as<float>(), e);
                QTTS_CHECK(ok, ERR_STATE, "codec decode: fused residual unit refused a shape it reported as supported");
                if (profiling) {
                    QTTS_HIP(hipEventRecord(pr.b, st));
                    prof.push_back(pr);
                }
                std::swap(in, outb);
            }
            std::swap(w.A, w.Bf);  // three units: the block's result sits in the other buffer; A names it again
        }
        for (int u = 0; u < 3 && !fuse && !f32in; ++u) {
            const CodecState::Unit& un = blk.u[u];
            // conv1 (k = 7, dilated): A -> snake(act2) in Bf
            gemm(un.c1, (const char*)w.A.p + (size_t)(Pn - 6 * un.c1.dil) * CPo * es, Mout, CPo);
            {
                ConvEpi e{};
                e.M = Mout; e.N = CPo; e.r = 1; e.Cout = CPo; e.bias = un.c1.bias.as<float>();
                e.lens = d_len(so); e.S_in = S[(size_t)so]; e.S_out = S[(size_t)so]; e.P_out = 0; e.ld_out = CPo;
                e.out2 = w.Bf.p; e.alpha_exp = un.a2.a.as<float>(); e.beta_den = un.a2.den.as<float>();
                epilogue(e);
            }
            // conv2 (k = 1) + residual: R += conv2(Bf); A = snake_next(R).  After the last unit the raw sum is dead.
            gemm(un.c2, w.Bf.p, Mout, CPo);
            {
                ConvEpi e{};
                e.M = Mout; e.N = CPo; e.r = 1; e.Cout = CPo; e.bias = un.c2.bias.as<float>();
                e.resid_f32 = w.Rf.as<float>(); e.lens = d_len(so); e.S_in = S[(size_t)so]; e.S_out = S[(size_t)so]; e.P_out = Pn; e.ld_out = CPo;
                const Snake& nx = u < 2 ? blk.u[u + 1].a1 : (b + 1 < nb ? s.blocks[(size_t)b + 1].s0 : s.sfinal);
                e.alpha_exp = nx.a.as<float>(); e.beta_den = nx.den.as<float>();
                e.out_f32 = u < 2 ? w.Rf.as<float>() : nullptr;  // after the last unit only the activated copy is read again
                e.out2 = w.A.p;
                e.out2_lo = (u == 2 && b + 1 < SPLIT_BLOCKS) ? lo : 0;
                epilogue(e);
            }
        }
        CP = CPo;
        if (b + 1 < nb) {
            // the next block overwrites A and R only after its transposed-conv GEMM has consumed xin, but its epilogue
            // writes A while nothing reads it any more; Bf takes over as the activation buffer to keep xin intact
            std::swap(w.A, w.Bf);
            xin = (const char*)w.Bf.p + (size_t)Pn * CP * es;
        }
    }
    // ================= final conv (C -> 1, k = 7) + clamp =================
    const int last = nstage - 1;
    if (!final_done)
        launch_final_conv(st, dt, w.A.p, last_narrow ? s.final_wn.as<float>() : s.final_w.as<float>(), s.final_b, d_len(last), U, g.Tmax[(size_t)last],
                          S[(size_t)last], P[(size_t)last], last_narrow ? 96 : s.final_CP, w.wav.as<float>(), g.wav_max);
    QTTS_HIP(hipEventRecord(ev1, st));
    const long up_total = c.total_upsample();
    for (int u = 0; u < U; ++u) {
        const Piece& p = ps[(size_t)u];
        // a streaming push re-decodes C context frames: everything up to the end of the previous push's output goes
        long drop = (long)p.drop_frames * up_total;
        if (sc && sc->C > 0) drop = (long)sc->C * up_total - ((long)Tm * up_total - L[(size_t)last][(size_t)u]);
        const long n = L[(size_t)last][(size_t)u] - drop;
        QTTS_CHECK(n >= 0, ERR_SHAPE, "codec decode: chunk shorter than its context");
        QTTS_CHECK(p.out_off + n <= wav_stride, ERR_SHAPE, "codec decode: wav_stride too small");
        if (n > 0)
            QTTS_HIP(hipMemcpyAsync(wav_out + (size_t)p.row * wav_stride + p.out_off, w.wav.as<float>() + (size_t)u * g.wav_max + drop,
                                    (size_t)n * 4, hipMemcpyDeviceToHost, st));
    }
    QTTS_HIP(hipStreamSynchronize(st));
    float ms = 0;
    QTTS_HIP(hipEventElapsedTime(&ms, ev0, ev1));
    m.timing[6] += ms;
    if (profiling) report();
}

}  // namespace

// Model::decode_output / TokenizerV2::{decode, chunked_decode} (model/generate.rs:29-143, audio/decoder/v2.rs:163-286).
void run_codec_decode(Model& m, const uint32_t* codes, const int* lens, int B, int T_max, int chunk, int left_ctx, float* wav_out,
                      int64_t wav_stride, int64_t* wav_lens) {
    QTTS_CHECK(m.finalized && m.codec_state, ERR_STATE, "codec weights are not loaded");
    QTTS_CHECK(B >= 1 && T_max >= 1 && wav_stride >= 1, ERR_BAD_ARG, "codec decode: bad batch shape");
    QTTS_CHECK(chunk == -2 || chunk <= 0 || left_ctx >= 0, ERR_BAD_ARG, "codec decode: negative left context");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    CodecState& s = *m.codec_state;
    CodecScratch& w = codec_scratch(m);
    const CodecCfg& c = m.codec;
    const long up_total = c.total_upsample();
    // every code id is validated on the host (quantizer lookups index device tables with them)
    for (int b = 0; b < B; ++b) {
        QTTS_CHECK(lens[b] >= 0 && lens[b] <= T_max, ERR_BAD_ARG, "codec decode: lens out of range");
        const uint32_t* row = codes + (size_t)b * T_max * s.Q;
        for (size_t i = 0; i < (size_t)lens[b] * s.Q; ++i)
            QTTS_CHECK(row[i] < (uint32_t)s.csize, ERR_BAD_ARG, "codec decode: code id out of range");
    }
    // ---- pieces
    std::vector<Piece> pieces;
    // samples one decode of n frames yields: every kernel == 2 * stride transposed conv drops one input step
    auto samples_of = [&](int n) {
        long L = n;
        for (int r : c.upsampling_ratios) L *= r;
        for (int r : c.upsample_rates) L = (L - 1) * r;
        return L;
    };
    for (int b = 0; b < B; ++b) {
        const int len = lens[b];
        wav_lens[b] = 0;
        if (len == 0) continue;
        int ch = chunk, ctx = left_ctx;
        if (chunk == -2) {  // decode_output policy (model/generate.rs:71-111)
            if (len > 512) { ch = 256; ctx = 32; } else ch = 0;
        }
        if (ch <= 0) {
            pieces.push_back(Piece{b, 0, len, 0, 0});
            wav_lens[b] = samples_of(len);
        } else {
            int64_t off = 0;
            for (int start = 0; start < len; start += ch) {
                const int end = std::min(start + ch, len);
                const int cx = start > ctx ? ctx : start;
                const long n = samples_of(end - start + cx) - (long)cx * up_total;
                QTTS_CHECK(n >= 0, ERR_SHAPE, "codec decode: chunk shorter than its context");
                pieces.push_back(Piece{b, start - cx, end - start + cx, cx, off});
                off += n;
            }
            wav_lens[b] = off;
        }
        QTTS_CHECK(wav_lens[b] <= wav_stride, ERR_SHAPE, "codec decode: wav_stride too small for the decoded length");
    }
    m.timing[6] = 0;
    if (pieces.empty()) return;
    std::stable_sort(pieces.begin(), pieces.end(), [](const Piece& a, const Piece& b) { return a.len > b.len; });
    w.codes.ensure((size_t)B * T_max * s.Q * 4);
    copy_sync(w.codes.p, codes, (size_t)B * T_max * s.Q * 4, hipMemcpyHostToDevice, m.ctx->stream);
    // ---- groups bounded by a scratch budget.  The vocoder's widest stage holds, per 12.5 Hz frame of an utterance, 1920 rows x 128
    // channels of two f32 streams plus the bf16 copies in front of it: about 3.6 MB per utterance-frame at the reference-default
    // shapes.  24 GiB per handle (18 utterances of 375 frames per group) costs 1 % of the decode's time against 48 GiB; 12 GiB 3.5 %.
    static const size_t budget = (size_t)(getenv("QTTS_CODEC_SCRATCH_MB") ? std::max(64, atoi(getenv("QTTS_CODEC_SCRATCH_MB"))) : 24576) << 20;
    Decoder dec(m);
    size_t i = 0;
    while (i < pieces.size()) {
        const int Tm = pieces[i].len;  // sorted: the first piece of a group is its longest
        size_t n = 1;
        while (i + n < pieces.size() && n < 4096 && make_geometry(c, s, (int)n + 1, Tm, m.dtype).bytes(dec.es) <= budget) ++n;
        std::vector<Piece> grp(pieces.begin() + (long)i, pieces.begin() + (long)(i + n));
        dec.run_group(w.codes.as<uint32_t>(), T_max, grp, wav_out, wav_stride);
        i += n;
    }
}

// ---- streaming decode -----------------------------------------------------------------------------------------------
// The decoder is causal, so the samples of a prefix never change when more frames arrive (a prefix of n frames decides
// samples [0, up * n - tail), tail = 555 at the default config).  The reference has no streaming decode; its chunked
// decode (audio/decoder/v2.rs:250-286) restarts the transformer and the convs with 32 frames of context and is inexact.
// Here the transformer continues from its ring of keys/values, and the conv stack re-decodes just enough context
// frames to cover its receptive field, so the concatenated pushes equal one whole decode.

// frames of transformer output the conv stack needs in front of a push so that the first sample it has to emit is exact
static int conv_context_frames(const CodecCfg& c) {
    long up = c.total_upsample(), L = 1;
    for (int r : c.upsampling_ratios) L *= r;
    for (int r : c.upsample_rates) L = (L - 1) * r;
    long need = (up - L) + 6;  // the first emitted sample sits `tail` samples before the context boundary; final conv k = 7
    for (size_t b = c.upsample_rates.size(); b-- > 0;) {
        need += 6 * (1 + 3 + 9);                                      // three residual units, k = 7, dilations 1, 3, 9
        need = (need + c.upsample_rates[b] - 1) / c.upsample_rates[b];  // transposed conv: output j reads inputs j / r and j / r + 1
    }
    need += 6;  // decoder.0, k = 7
    for (size_t i = c.upsampling_ratios.size(); i-- > 0;) {
        need += 6;  // ConvNeXt depthwise k = 7
        need = (need + c.upsampling_ratios[i] - 1) / c.upsampling_ratios[i];
    }
    return (int)need + 2;  // + pre_conv's two frames of code history
}

CodecStream* codec_stream_open(Model& m, int B, int max_chunk) {
    QTTS_CHECK(m.finalized && m.codec_state, ERR_STATE, "codec weights are not loaded");
    QTTS_CHECK(B >= 1 && B <= 4096 && max_chunk >= 1 && max_chunk <= 8192, ERR_BAD_ARG, "codec stream: bad batch or chunk size");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    const CodecCfg& c = m.codec;
    CodecState& s = *m.codec_state;
    std::unique_ptr<CodecStream> sc(new CodecStream());
    sc->m = &m;
    sc->B = B;
    sc->max_chunk = max_chunk;
    sc->ctx_cap = conv_context_frames(c);
    // the ring must hold the window of the oldest token of a push while the newest one is written
    int pages = 1;
    while ((long)pages * PAGE < (long)sc->ctx_cap + max_chunk + c.window + PAGE) pages *= 2;
    sc->ring_pages = pages;
    const size_t es = dtype_size(m.dtype);
    const size_t page_elems = (size_t)2 * c.kv_heads * PAGE * c.head_dim;
    CodecScratch& w = codec_scratch(m);
    const size_t kv_bytes = (size_t)c.layers * ((size_t)B + 1) * pages * page_elems * es;
    const size_t lat_bytes = (size_t)B * sc->ctx_cap * s.latP * es * 2;  // room for the remainder plane of bf16 runs
    // (neither buffer is read before it is written: a fresh allocation and a handed-down one serve alike)
    if (w.spare_kv.bytes >= kv_bytes) sc->kv = std::move(w.spare_kv);
    else sc->kv.alloc(kv_bytes);
    if (w.spare_latent.bytes >= lat_bytes) sc->latent_ctx = std::move(w.spare_latent);
    else sc->latent_ctx.alloc(lat_bytes);
    return sc.release();
}

void codec_stream_close(CodecStream* sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->m->ctx->device);
    (void)hipStreamSynchronize(sc->m->ctx->stream);
    if (sc->m->codec_scratch) {
        CodecScratch& w = *sc->m->codec_scratch;
        if (sc->kv.bytes > w.spare_kv.bytes) std::swap(sc->kv, w.spare_kv);
        if (sc->latent_ctx.bytes > w.spare_latent.bytes) std::swap(sc->latent_ctx, w.spare_latent);
    }
    delete sc;
}

// codes [B][n][Q]: the next n frames of every row.  wav_out [B][wav_stride] receives the samples these frames decide:
// up * n - tail for the first push, up * n afterwards; *n_samples is that count.
void codec_stream_push(CodecStream& sc, const uint32_t* codes, int n, float* wav_out, int64_t wav_stride, int64_t* n_samples) {
    Model& m = *sc.m;
    QTTS_CHECK(m.finalized && m.codec_state, ERR_STATE, "codec weights are not loaded");
    QTTS_CHECK(n >= 1 && n <= sc.max_chunk, ERR_BAD_ARG, "codec stream: push is empty or longer than the session's max_chunk");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    CodecState& s = *m.codec_state;
    CodecScratch& w = codec_scratch(m);
    const CodecCfg& c = m.codec;
    const int B = sc.B, Q = s.Q, C = sc.C, T = C + n;
    for (size_t i = 0; i < (size_t)B * n * Q; ++i) QTTS_CHECK(codes[i] < (uint32_t)s.csize, ERR_BAD_ARG, "codec stream: code id out of range");
    long L = T;
    for (int r : c.upsampling_ratios) L *= r;
    for (int r : c.upsample_rates) L = (L - 1) * r;
    const long up = c.total_upsample();
    const long out = C > 0 ? (long)n * up : L;
    QTTS_CHECK(out >= 1, ERR_SHAPE, "codec stream: the first push is too short to decide any sample");
    QTTS_CHECK(out <= wav_stride, ERR_SHAPE, "codec stream: wav_stride too small");
    QTTS_CHECK(make_geometry(c, s, B, T, m.dtype).bytes(dtype_size(m.dtype)) <= ((size_t)96 << 30), ERR_OOM,
               "codec stream: batch x chunk needs more than 96 GB of scratch; push fewer frames at a time");
    // context codes in front of the new ones
    std::vector<uint32_t> all((size_t)B * T * Q);
    for (int b = 0; b < B; ++b) {
        uint32_t* dst = &all[(size_t)b * T * Q];
        if (C > 0) std::memcpy(dst, &sc.ctx_codes[(size_t)b * C * Q], (size_t)C * Q * 4);
        std::memcpy(dst + (size_t)C * Q, codes + (size_t)b * n * Q, (size_t)n * Q * 4);
    }
    w.codes.ensure(all.size() * 4);
    copy_sync(w.codes.p, all.data(), all.size() * 4, hipMemcpyHostToDevice, m.ctx->stream);
    std::vector<Piece> ps;
    for (int b = 0; b < B; ++b) ps.push_back(Piece{b, 0, T, 0, 0});
    m.timing[6] = 0;
    Decoder dec(m);
    dec.run_group(w.codes.as<uint32_t>(), T, ps, wav_out, wav_stride, &sc);
    // session state for the next push
    const int Cn = std::min(T, sc.ctx_cap);
    std::vector<uint32_t> ctx((size_t)B * Cn * Q);
    for (int b = 0; b < B; ++b) std::memcpy(&ctx[(size_t)b * Cn * Q], &all[((size_t)b * T + (T - Cn)) * Q], (size_t)Cn * Q * 4);
    sc.ctx_codes.swap(ctx);
    sc.C = Cn;
    sc.frames += n;
    *n_samples = out;
}

}  // namespace qtts
