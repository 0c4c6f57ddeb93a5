// 12.5 Hz codec ENCODER on the GPU: 24 kHz waveform -> codes (qwen_tts/src/audio/encoder/v2.rs:87-120).
//
// SeaNet front (seanet.rs:153-176), 8-layer causal transformer (transformer.rs:170-184), stride-2 downsample
// (seanet.rs:243-249) and the split residual vector quantizer (quantizer.rs:117-179).  Activations are channels-last
// strips like in the decoder.  A strided conv with kernel 2r and stride r is a two-tap GEMM over the same buffer viewed
// as rows of r consecutive time steps (the mirror image of the decoder's transposed convs).
#include <algorithm>
#include <cmath>

#include "generate.hpp"
#include "prep.hpp"
#include "runner.hpp"

namespace qtts {

namespace {
constexpr int ENC_DIM = 512, ENC_HEADS = 8, ENC_HD = 64, ENC_MLP = 2048, ENC_LAYERS = 8, ENC_BINS = 2048, ENC_CD = 256;
constexpr int ENC_RATIOS[4] = {4, 5, 6, 8};
constexpr int ENC_PAGE = 64;
constexpr long ENC_SLACK = (CONV_HALO_SLACK + 64) * 8;  // fine rows readable past the last strip (the stride-8 view reads whole groups)

// conv (Cout, Cin, 2r) with stride r -> W'[co][tap * (r * CinP) + jj * CinP + ci] = w[co][ci][tap * r + jj]
void strided_conv_weight(const Model& m, const std::string& prefix, int r, bool bias, GemmW& g) {
    const Tensor& t = m.get(prefix + ".weight");
    QTTS_CHECK(t.shape.size() == 3 && t.shape[2] == 2 * r, ERR_SHAPE, "encoder: strided conv must have kernel == 2 * stride: " + prefix);
    const int Cout = (int)t.shape[0], Cin = (int)t.shape[1];
    const std::vector<float> w = host_f32(m, prefix + ".weight");
    g.CinP = pad64(Cin);
    g.CoutP = pad64(Cout);
    g.taps = 2;
    g.dil = 1;
    g.r = 1;
    g.N = g.CoutP;
    g.K = 2 * r * g.CinP;
    std::vector<float> out((size_t)g.N * g.K, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int k = 0; k < 2 * r; ++k) out[(size_t)co * g.K + (size_t)k * g.CinP + ci] = w[((size_t)co * Cin + ci) * 2 * r + k];
    upload(g.w, out, m.dtype);
    load_bias(m, bias ? prefix + ".bias" : std::string(), g.CoutP, g);
}
}  // namespace

struct EncoderState {
    DevBuf init_w, init_b;
    struct Stage {
        GemmW c1, c2, down;
        int C = 0, r = 0;
    } stage[4];
    GemmW final_conv, ds;
    struct Layer {
        DevBuf qkv;
        const void *o = nullptr, *fc1 = nullptr, *fc2 = nullptr;
        const float *ln1w = nullptr, *ln1b = nullptr, *ln2w = nullptr, *ln2b = nullptr, *ls_a = nullptr, *ls_m = nullptr;
    } layer[ENC_LAYERS];
    DevBuf cos, sin;
    int max_pos = 0;
    struct Codebook {
        DevBuf emb_t, emb, c2;  // [bins][cd] model dtype (GEMM operand) | f32 | |e|^2 / 2
    };
    struct Group {
        GemmW in_proj;
        std::vector<Codebook> layers;
    } group[2];
    // run-time buffers
    Workspace ws;
    DevBuf audio, nsamp, meta, X, A, Hh, D, acc, res, res_t, kv, toks, tiles, codes;
};

void finalize_encoder(Model& m) {
    if (!m.has("encoder.encoder.layers.0.conv.weight")) return;
    QTTS_HIP(hipSetDevice(m.ctx->device));
    auto st = std::make_shared<EncoderState>();
    EncoderState& s = *st;
    const std::string E = "encoder.encoder.layers.";
    {
        const Tensor& t = m.get(E + "0.conv.weight");
        QTTS_CHECK(t.shape == std::vector<int64_t>({64, 1, 7}), ERR_SHAPE, "encoder: input conv must be (64, 1, 7)");
        upload(s.init_w, host_f32(m, E + "0.conv.weight"), F32);
        upload(s.init_b, host_f32(m, E + "0.conv.bias"), F32);
    }
    int C = 64, idx = 1;
    for (int i = 0; i < 4; ++i) {
        EncoderState::Stage& g = s.stage[i];
        g.C = C;
        g.r = ENC_RATIOS[i];
        conv_weight(m, E + std::to_string(idx) + ".block.1.conv", 1, g.c1);
        conv_weight(m, E + std::to_string(idx) + ".block.3.conv", 1, g.c2);
        strided_conv_weight(m, E + std::to_string(idx + 2) + ".conv", g.r, true, g.down);
        QTTS_CHECK(g.c1.taps == 3 && g.c1.CinP == pad64(C) && g.c2.taps == 1 && g.c2.CinP == g.c1.CoutP && g.c2.CoutP == pad64(C) &&
                       g.down.CinP == pad64(C) && g.down.CoutP == pad64(2 * C),
                   ERR_SHAPE, "encoder: SeaNet stage shapes do not chain");
        C *= 2;
        idx += 3;
    }
    conv_weight(m, E + std::to_string(idx + 1) + ".conv", 1, s.final_conv);
    QTTS_CHECK(s.final_conv.taps == 3 && s.final_conv.CinP == C && s.final_conv.CoutP == ENC_DIM, ERR_SHAPE, "encoder: final conv must be (512, 1024, 3)");
    const size_t es = dtype_size(m.dtype);
    hipStream_t stream = m.ctx->stream;
    for (int l = 0; l < ENC_LAYERS; ++l) {
        const std::string L = "encoder.encoder_transformer.layers." + std::to_string(l) + ".";
        EncoderState::Layer& y = s.layer[l];
        for (const char* n : {"q", "k", "v", "o"}) {
            const Tensor& t = m.get(L + "self_attn." + n + "_proj.weight");
            QTTS_CHECK(t.shape == std::vector<int64_t>({ENC_DIM, ENC_DIM}), ERR_SHAPE, "encoder: attention projections must be (512, 512)");
        }
        y.qkv.alloc((size_t)3 * ENC_DIM * ENC_DIM * es);
        const size_t one = (size_t)ENC_DIM * ENC_DIM * es;
        QTTS_HIP(hipMemcpyAsync(y.qkv.p, m.mat(L + "self_attn.q_proj.weight"), one, hipMemcpyDeviceToDevice, stream));
        QTTS_HIP(hipMemcpyAsync((char*)y.qkv.p + one, m.mat(L + "self_attn.k_proj.weight"), one, hipMemcpyDeviceToDevice, stream));
        QTTS_HIP(hipMemcpyAsync((char*)y.qkv.p + 2 * one, m.mat(L + "self_attn.v_proj.weight"), one, hipMemcpyDeviceToDevice, stream));
        y.o = m.mat(L + "self_attn.o_proj.weight");
        y.fc1 = m.mat(L + "mlp.fc1.weight");
        y.fc2 = m.mat(L + "mlp.fc2.weight");
        QTTS_CHECK(m.get(L + "mlp.fc1.weight").shape == std::vector<int64_t>({ENC_MLP, ENC_DIM}) &&
                       m.get(L + "mlp.fc2.weight").shape == std::vector<int64_t>({ENC_DIM, ENC_MLP}),
                   ERR_SHAPE, "encoder: MLP must be 512 -> 2048 -> 512");
        y.ln1w = m.f32(L + "input_layernorm.weight");
        y.ln1b = m.f32(L + "input_layernorm.bias");
        y.ln2w = m.f32(L + "post_attention_layernorm.weight");
        y.ln2b = m.f32(L + "post_attention_layernorm.bias");
        y.ls_a = m.f32(L + "self_attn_layer_scale.scale");
        y.ls_m = m.f32(L + "mlp_layer_scale.scale");
    }
    QTTS_HIP(hipStreamSynchronize(stream));
    strided_conv_weight(m, "encoder.downsample.conv", 2, false, s.ds);
    QTTS_CHECK(s.ds.CinP == ENC_DIM && s.ds.CoutP == ENC_DIM, ERR_SHAPE, "encoder: downsample conv must be (512, 512, 4)");
    for (int gi = 0; gi < 2; ++gi) {
        const std::string P = gi == 0 ? "encoder.quantizer.semantic_residual_vector_quantizer." : "encoder.quantizer.acoustic_residual_vector_quantizer.";
        EncoderState::Group& g = s.group[gi];
        {   // input_proj (256, 512, 1) as a linear layer
            const Tensor& t = m.get(P + "input_proj.weight");
            QTTS_CHECK(t.shape.size() == 3 && t.shape[0] == ENC_CD && t.shape[1] == ENC_DIM && t.shape[2] == 1, ERR_SHAPE, "encoder: quantizer input_proj must be (256, 512, 1)");
            upload(g.in_proj.w, host_f32(m, P + "input_proj.weight"), m.dtype);
            g.in_proj.N = ENC_CD; g.in_proj.K = ENC_DIM; g.in_proj.CinP = ENC_DIM; g.in_proj.CoutP = ENC_CD;
        }
        const int want = gi == 0 ? 1 : 31;
        for (int l = 0; l < want; ++l) {
            const std::string Cb = P + "layers." + std::to_string(l) + ".codebook.";
            if (!m.has(Cb + "embed_sum")) break;  // only the first valid_num_quantizers - 1 acoustic layers are ever used
            const std::vector<float> sum = host_f32(m, Cb + "embed_sum"), usage = host_f32(m, Cb + "cluster_usage");
            QTTS_CHECK((int)usage.size() == ENC_BINS && (int)sum.size() == ENC_BINS * ENC_CD, ERR_SHAPE, "encoder: codebook must be (2048, 256)");
            std::vector<float> emb(sum.size()), c2((size_t)ENC_BINS);
            for (int r = 0; r < ENC_BINS; ++r) {  // EncoderCodebook::embeddings / c2 (quantizer.rs:23-37)
                const float u = std::max(usage[(size_t)r], 1e-5f);
                float s2 = 0.f;
                for (int i = 0; i < ENC_CD; ++i) {
                    const float e = sum[(size_t)r * ENC_CD + i] / u;
                    emb[(size_t)r * ENC_CD + i] = e;
                    s2 += e * e;
                }
                c2[(size_t)r] = s2 * 0.5f;
            }
            g.layers.emplace_back();
            upload(g.layers.back().emb, emb, F32);
            upload(g.layers.back().emb_t, emb, m.dtype);
            upload(g.layers.back().c2, c2, F32);
        }
        QTTS_CHECK(!g.layers.empty(), ERR_STATE, "encoder: quantizer group without codebooks");
    }
    m.enc_state = st;
}

// TokenizerV2::encode for a batch of clips.  codes_out [B][T_cap][Q] u32, T_out[b] = frames of clip b.
void run_encode_audio(Model& m, const float* audio, const int* n_samples, int B, int max_samples, int Q, uint32_t* codes_out, int T_cap,
                      int* T_out) {
    std::lock_guard<std::mutex> lock(m.locks->encoder);  // sessions of one model share these workspaces
    QTTS_CHECK(m.finalized && m.enc_state, ERR_STATE, "Encoder not loaded. Load the speech tokenizer's encoder.* tensors to encode audio.");
    QTTS_CHECK(B >= 1 && max_samples >= 1 && Q >= 1 && T_cap >= 1, ERR_BAD_ARG, "encode_audio: bad batch shape");
    EncoderState& s = *m.enc_state;
    QTTS_CHECK(Q <= 1 + (int)s.group[1].layers.size(), ERR_BAD_ARG, "encode_audio: more quantizers requested than codebooks loaded");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    hipStream_t st = m.ctx->stream;
    const int dt = m.dtype;
    const size_t es = dtype_size(dt);
    // groups of clips bounded by a scratch budget (the first SeaNet stage holds four 64-channel tensors at 24 kHz)
    const size_t per_clip = (size_t)max_samples * 64 * es * 4 + ((size_t)1 << 20);
    const int group_max = (int)std::max<size_t>(1, std::min<size_t>(64, ((size_t)8 << 30) / per_clip));
    for (int b0 = 0; b0 < B; b0 += group_max) {
        const int U = std::min(group_max, B - b0);
        // ---- lengths per stage: 0 = samples, 1..4 after each strided conv, 5 after the stride-2 downsample
        std::vector<std::vector<int>> L(6, std::vector<int>((size_t)U));
        long Tmax[6] = {0, 0, 0, 0, 0, 0};
        for (int u = 0; u < U; ++u) {
            int n = n_samples[b0 + u];
            QTTS_CHECK(n >= 1 && n <= max_samples, ERR_BAD_ARG, "encode_audio: n_samples out of range");
            L[0][(size_t)u] = n;
            for (int i = 0; i < 4; ++i) { n = n / ENC_RATIOS[i]; L[(size_t)i + 1][(size_t)u] = n; }
            QTTS_CHECK(n >= 2, ERR_BAD_ARG, "encode_audio: clip shorter than one code frame (1920 samples)");
            L[5][(size_t)u] = n / 2;
            for (int i = 0; i < 6; ++i) Tmax[i] = std::max<long>(Tmax[i], L[(size_t)i][(size_t)u]);
            QTTS_CHECK(L[5][(size_t)u] <= T_cap, ERR_SHAPE, "encode_audio: T_cap too small");
            T_out[b0 + u] = L[5][(size_t)u];
        }
        // strip geometry: stages 0..3 feed a strided conv (P = r, S a multiple of r with one spare group), stage 4 feeds the
        // final k = 3 conv (P = 2), the transformer rows have no padding, its output is re-stripped with P = 2 for the downsample
        long S[5];
        int P[5];
        for (int i = 0; i < 4; ++i) {
            const int r = ENC_RATIOS[i];
            P[i] = r;
            S[i] = ((r + Tmax[i] + r - 1) / r + 1) * r;
        }
        P[4] = 2;
        S[4] = 2 + Tmax[4];
        const long Sx = Tmax[4];                 // transformer rows per strip
        const long Sd = ((2 + Tmax[4] + 1) / 2 + 1) * 2;  // downsample input strips (P = 2)
        const long Sq = Sd / 2;                  // rows per strip of the downsample GEMM / quantizer
        // ---- metadata
        std::vector<int> meta;
        size_t o_len[6];
        for (int i = 0; i < 6; ++i) { o_len[i] = meta.size(); meta.insert(meta.end(), L[(size_t)i].begin(), L[(size_t)i].end()); }
        s.meta.ensure(meta.size() * 4);
        QTTS_HIP(hipMemcpyAsync(s.meta.p, meta.data(), meta.size() * 4, hipMemcpyHostToDevice, st));
        const int* d_meta = s.meta.as<int>();
        auto d_len = [&](int i) { return d_meta + o_len[i]; };
        s.audio.ensure((size_t)U * max_samples * 4);
        s.nsamp.ensure((size_t)U * 4);
        QTTS_HIP(hipMemcpyAsync(s.audio.p, audio + (size_t)b0 * max_samples, (size_t)U * max_samples * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipMemcpyAsync(s.nsamp.p, n_samples + b0, (size_t)U * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipStreamSynchronize(st));  // `meta` is a local
        // ---- buffers
        size_t act = 0;
        for (int i = 0; i < 4; ++i) act = std::max(act, (size_t)(U * S[i] + ENC_SLACK) * (size_t)pad64(s.stage[i].C));
        act = std::max(act, (size_t)(U * S[4] + ENC_SLACK) * 1024);
        act = std::max(act, (size_t)(U * Sd + ENC_SLACK) * ENC_DIM);
        s.X.ensure(act * es);
        s.A.ensure(act * es);
        s.Hh.ensure(act * es);
        s.D.ensure(act * es);
        const long Mq = U * Sq;
        s.acc.ensure((size_t)((Mq + 63) / 64 * 64) * ENC_BINS * 4);
        s.res.ensure((size_t)(Mq + ENC_SLACK) * ENC_CD * 4);
        s.res_t.ensure((size_t)(Mq + ENC_SLACK) * ENC_CD * es);
        s.codes.ensure((size_t)U * T_cap * Q * 4);
        QTTS_HIP(hipMemsetAsync(s.codes.p, 0, (size_t)U * T_cap * Q * 4, st));
        auto zero = [&](DevBuf& d, long rows, int C) { QTTS_HIP(hipMemsetAsync(d.p, 0, (size_t)(rows + ENC_SLACK) * C * es, st)); };
        auto conv = [&](const GemmW& g, const void* X, int ldx, long M, ConvEpi e) {
            e.M = M; e.N = g.N; e.r = 1; e.Cout = g.CoutP; e.bias = g.bias.p ? g.bias.as<float>() : nullptr;
            launch_gemm_rows_epi(st, dt, g.wm(), X, g.K, ldx, g.taps, (long)g.dil * ldx, e);
        };
        auto at = [&](const DevBuf& d, long row, int ld) { return (const char*)d.p + (size_t)row * ld * es; };

        // ================= SeaNet =================
        zero(s.X, U * S[0], 64);
        zero(s.A, U * S[0], 64);
        launch_enc_init_conv(st, dt, s.audio.as<float>(), s.nsamp.as<int>(), U, max_samples, s.init_w.as<float>(), s.init_b.as<float>(), S[0], P[0],
                             s.X.p, s.A.p);
        for (int i = 0; i < 4; ++i) {
            const EncoderState::Stage& g = s.stage[i];
            const int CP = pad64(g.C), HP = g.c1.CoutP, r = g.r;
            const long M = U * S[i];
            // residual block: conv1 (k = 3) on ELU(x) -> ELU -> conv2 (k = 1) -> + x, then ELU for the strided conv
            zero(s.Hh, M, HP);
            {
                ConvEpi e{};
                e.act = 4; e.lens = d_len(i); e.S_in = S[i]; e.S_out = S[i]; e.P_out = 0; e.ld_out = HP; e.out = s.Hh.p;
                conv(g.c1, at(s.A, P[i] - 2, CP), CP, M, e);
            }
            zero(s.D, M, CP);
            {
                ConvEpi e{};
                e.resid = s.X.p; e.lens = d_len(i); e.S_in = S[i]; e.S_out = S[i]; e.P_out = P[i]; e.ld_out = CP; e.out2 = s.D.p; e.act2 = 1;
                conv(g.c2, s.Hh.p, HP, M, e);
            }
            // strided conv: the buffer viewed as rows of r time steps; output step t reads groups t and t + 1
            const long Mg = U * (S[i] / r);
            const int CoP = g.down.CoutP;
            if (i < 3) {
                zero(s.X, U * S[i + 1], CoP);
                zero(s.A, U * S[i + 1], CoP);
                ConvEpi e{};
                e.lens = d_len(i + 1); e.S_in = S[i] / r; e.S_out = S[i + 1]; e.P_out = P[i + 1]; e.ld_out = CoP; e.out = s.X.p; e.out2 = s.A.p; e.act2 = 1;
                conv(g.down, s.D.p, r * CP, Mg, e);
            } else {
                zero(s.A, U * S[4], CoP);
                ConvEpi e{};
                e.lens = d_len(4); e.S_in = S[i] / r; e.S_out = S[4]; e.P_out = P[4]; e.ld_out = CoP; e.out2 = s.A.p; e.act2 = 1;
                conv(g.down, s.D.p, r * CP, Mg, e);
            }
        }
        // ================= transformer over every strip row =================
        const long Mx = U * Sx;
        QTTS_CHECK(Mx < (1L << 30), ERR_SHAPE, "encode_audio: group too large");
        const int M = (int)Mx;
        const size_t rp = (size_t)gemm_pad_rows(M) + 64;
        s.ws.x.ensure(rp * ENC_DIM * 4);
        s.ws.xn.ensure(rp * ENC_DIM * es);
        s.ws.part.ensure((size_t)ENC_MLP * std::max(rp, (size_t)GEMM_MAX_KS * std::min<size_t>(rp, 512)) * 4);
        s.ws.q.ensure(rp * ENC_DIM * 4);
        s.ws.attn.ensure(rp * ENC_DIM * es);
        s.ws.act.ensure(rp * ENC_MLP * es);
        QTTS_HIP(hipMemsetAsync(s.ws.x.p, 0, rp * ENC_DIM * 4, st));
        {   // final SeaNet conv (k = 3) writes the f32 residual stream of the transformer
            ConvEpi e{};
            e.lens = d_len(4); e.S_in = S[4]; e.S_out = Sx; e.P_out = 0; e.ld_out = ENC_DIM; e.out_f32 = s.ws.x.as<float>();
            conv(s.final_conv, at(s.A, P[4] - 2, 1024), 1024, U * S[4], e);
        }
        const int pp = (int)((Tmax[4] + ENC_PAGE - 1) / ENC_PAGE);
        std::vector<Tok> toks((size_t)M);
        for (int u = 0; u < U; ++u)
            for (long r = 0; r < Sx; ++r) {
                Tok& t = toks[(size_t)(u * Sx + r)];
                if (r < L[4][(size_t)u]) t = Tok{u, (int)r, 0, (int)r + 1};  // causal, no window (transformer.rs:88-96)
                else { const int slot = (int)((u * Sx + r) % ENC_PAGE); t = Tok{U, slot, slot, 1}; }
            }
        s.toks.ensure(toks.size() * sizeof(Tok));
        QTTS_HIP(hipMemcpyAsync(s.toks.p, toks.data(), toks.size() * sizeof(Tok), hipMemcpyHostToDevice, st));
        std::vector<AttnTile> tiles;
        build_attn_tiles(toks.data(), M, tiles);
        s.tiles.ensure(tiles.size() * sizeof(AttnTile));
        QTTS_HIP(hipMemcpyAsync(s.tiles.p, tiles.data(), tiles.size() * sizeof(AttnTile), hipMemcpyHostToDevice, st));
        QTTS_HIP(hipStreamSynchronize(st));
        const size_t page_elems = (size_t)2 * ENC_HEADS * ENC_PAGE * ENC_HD;
        s.kv.ensure(((size_t)U + 1) * pp * page_elems * es);
        KVView kv{s.kv.p, 0, ENC_PAGE, ENC_HEADS, ENC_HD, pp, nullptr};
        if (s.max_pos < std::max<long>(Tmax[4], ENC_PAGE)) {
            int n = 256;
            while (n < Tmax[4]) n *= 2;
            std::vector<float> cs, sn;
            rope_table_simple(ENC_HD, 10000.0, n, cs, sn);
            s.cos.alloc(cs.size() * 4);
            s.sin.alloc(sn.size() * 4);
            copy_sync(s.cos.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, m.ctx->stream);
            copy_sync(s.sin.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, m.ctx->stream);
            s.max_pos = n;
        }
        Runner run(m, s.ws, nullptr);
        run.tiles = s.tiles.as<AttnTile>();
        run.n_tiles = (int)tiles.size();
        run.split(false);
        StackCfg acfg{};
        acfg.d = ENC_HD; acfg.nh = ENC_HEADS; acfg.nkv = ENC_HEADS;
        float* x = s.ws.x.as<float>();
        float* part = s.ws.part.as<float>();
        launch_resid_layernorm(st, dt, x, nullptr, 0, 0, nullptr, s.layer[0].ln1w, s.layer[0].ln1b, 1e-5f, x, s.ws.xn.p, M, ENC_DIM);
        for (int l = 0; l < ENC_LAYERS; ++l) {
            const EncoderState::Layer& y = s.layer[l];
            int ks = run.ks_for(M, 3 * ENC_DIM, ENC_DIM);
            run.gemm(y.qkv.p, s.ws.xn.p, M, 3 * ENC_DIM, ENC_DIM, ks, ENC_DIM);
            QkvPostArgs qa{};
            qa.part = part; qa.KS = ks; qa.ks_stride = run.kstride(M, 3 * ENC_DIM); qa.eps = 1e-5f; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
            qa.toks = s.toks.as<Tok>(); qa.M = M; qa.nh = ENC_HEADS; qa.nkv = ENC_HEADS; qa.d = ENC_HD; qa.layer = 0; qa.q_out = s.ws.q.as<float>();
            launch_qkv_post(st, dt, qa, kv);
            run.attention(acfg, s.toks.as<Tok>(), kv, 0, M, 1.0f / std::sqrt((float)ENC_HD));
            ks = run.ks_for(M, ENC_DIM, ENC_DIM);
            run.gemm(y.o, s.ws.attn.p, M, ENC_DIM, ENC_DIM, ks, ENC_DIM);
            launch_resid_layernorm(st, dt, x, part, ks, run.kstride(M, ENC_DIM), y.ls_a, y.ln2w, y.ln2b, 1e-5f, x, s.ws.xn.p, M, ENC_DIM);
            ks = run.ks_for(M, ENC_MLP, ENC_DIM);
            run.gemm(y.fc1, s.ws.xn.p, M, ENC_MLP, ENC_DIM, ks, ENC_DIM);
            launch_bias_act(st, part, ks, run.kstride(M, ENC_MLP), nullptr, 2, s.ws.act.p, dt, M, ENC_MLP);
            ks = run.ks_for(M, ENC_DIM, ENC_MLP);
            run.gemm(y.fc2, s.ws.act.p, M, ENC_DIM, ENC_MLP, ks, ENC_MLP);
            const bool last = l + 1 == ENC_LAYERS;
            launch_resid_layernorm(st, dt, x, part, ks, run.kstride(M, ENC_DIM), y.ls_m, last ? nullptr : s.layer[l + 1].ln1w,
                                   last ? nullptr : s.layer[l + 1].ln1b, 1e-5f, x, s.ws.xn.p, M, ENC_DIM);
        }
        // ================= downsample (k = 4, stride 2, no bias) =================
        zero(s.D, U * Sd, ENC_DIM);
        launch_rows_to_strips(st, dt, x, Sx, 0, d_len(4), U, (int)Tmax[4], ENC_DIM, s.D.p, Sd, 2);
        zero(s.X, Mq, ENC_DIM);
        {
            ConvEpi e{};
            e.lens = d_len(5); e.S_in = Sq; e.S_out = Sq; e.P_out = 0; e.ld_out = ENC_DIM; e.out = s.X.p;
            conv(s.ds, s.D.p, 2 * ENC_DIM, Mq, e);
        }
        // ================= split residual vector quantizer =================
        int q = 0;
        for (int gi = 0; gi < 2 && q < Q; ++gi) {
            const EncoderState::Group& g = s.group[gi];
            launch_gemm_rows(st, dt, g.in_proj.wm(), s.X.p, s.acc.as<float>(), Mq, ENC_CD, ENC_DIM, ENC_DIM, 1, 0);
            QTTS_HIP(hipMemcpyAsync(s.res.p, s.acc.p, (size_t)Mq * ENC_CD * 4, hipMemcpyDeviceToDevice, st));
            launch_cast_rows(st, s.acc.as<float>(), s.res_t.p, dt, Mq * ENC_CD);
            for (size_t l = 0; l < g.layers.size() && q < Q; ++l, ++q) {
                const EncoderState::Codebook& cb = g.layers[l];
                launch_gemm_rows(st, dt, cb.emb_t.p, s.res_t.p, s.acc.as<float>(), Mq, ENC_BINS, ENC_CD, ENC_CD, 1, 0);
                launch_vq_argmin(st, dt, s.acc.as<float>(), cb.c2.as<float>(), cb.emb.as<float>(), ENC_BINS, ENC_CD, s.res.as<float>(), s.res_t.p,
                                 d_len(5), Sq, U, s.codes.as<unsigned>(), T_cap, Q, q);
            }
        }
        QTTS_HIP(hipMemcpyAsync(codes_out + (size_t)b0 * T_cap * Q, s.codes.p, (size_t)U * T_cap * Q * 4, hipMemcpyDeviceToHost, st));
        QTTS_HIP(hipStreamSynchronize(st));
    }
}

}  // namespace qtts
