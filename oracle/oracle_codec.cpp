// CPU oracle (test infrastructure): 12.5 Hz codec decoder (codes -> waveform).
// Restates qwen_tts/src/audio/decoder/v2.rs:163-286 and the modules it calls.
// Activations are channels-first [C][T] per utterance, as in the reference.
#include "oracle_codec.hpp"

#include <algorithm>
#include <stdexcept>

namespace orc {

// CausalConv1d (audio/tokenizer/v2/causal_conv.rs:15-101): left zero-pad (k-1)*dil (stride 1),
// Candle conv1d weight layout (Cout, Cin/groups, k), bias.
static void causal_conv1d(const std::vector<float>& x, int Cin, int T, const Tensor& w, const float* b, int dil,
                          int groups, std::vector<float>& y) {
    const int Cout = (int)w.dim(0), cig = (int)w.dim(1), k = (int)w.dim(2);
    if (cig * groups != Cin) throw std::runtime_error("oracle: conv1d channel mismatch");
    const int pad = (k - 1) * dil;
    const int cog = Cout / groups;
    y.assign((size_t)Cout * T, 0.f);
#pragma omp parallel
    {
        std::vector<float> patch((size_t)cig * k);
#pragma omp for schedule(static)
        for (int t = 0; t < T; ++t) {
            for (int g = 0; g < groups; ++g) {
                for (int ci = 0; ci < cig; ++ci)
                    for (int kk = 0; kk < k; ++kk) {
                        int xi = t + kk * dil - pad;
                        patch[(size_t)ci * k + kk] = xi >= 0 ? x[(size_t)(g * cig + ci) * T + xi] : 0.f;
                    }
                for (int co = g * cog; co < (g + 1) * cog; ++co) {
                    float s = dot(patch.data(), w.p() + (size_t)co * cig * k, (int64_t)cig * k);
                    if (b) s += b[co];
                    y[(size_t)co * T + t] = s;
                }
            }
        }
    }
}

// CausalConvTranspose1d (causal_conv.rs:107-176): Candle conv_transpose1d(stride s, kernel k, weight
// (Cin, Cout, k)) then drop k-s samples from BOTH ends.
static void causal_conv_transpose1d(const std::vector<float>& x, int Cin, int T, const Tensor& w, const float* b, int s,
                                    std::vector<float>& y, int& Tout) {
    const int Cout = (int)w.dim(1), k = (int)w.dim(2);
    if ((int)w.dim(0) != Cin) throw std::runtime_error("oracle: convT channel mismatch");
    const int full = (T - 1) * s + k;
    const int trim = k > s ? k - s : 0;
    Tout = full - 2 * trim;
    if (Tout <= 0) throw std::runtime_error("oracle: convT output empty");
    y.assign((size_t)Cout * Tout, 0.f);
#pragma omp parallel
    {
        std::vector<float> col((size_t)Cin), wcol((size_t)Cin);
#pragma omp for schedule(static)
        for (int co = 0; co < Cout; ++co) {
            for (int j0 = 0; j0 < Tout; ++j0) {
                const int j = j0 + trim;  // index in the untrimmed output
                float acc = b ? b[co] : 0.f;
                // contributions: j = t*s + kk, 0 <= kk < k, ascending t
                for (int t = std::max(0, (j - k + s) / s); t <= j / s && t < T; ++t) {
                    int kk = j - t * s;
                    if (kk < 0 || kk >= k) continue;
                    for (int ci = 0; ci < Cin; ++ci) {
                        col[(size_t)ci] = x[(size_t)ci * T + t];
                        wcol[(size_t)ci] = w.v[((size_t)ci * Cout + co) * k + kk];
                    }
                    acc += dot(col.data(), wcol.data(), Cin);
                }
                y[(size_t)co * Tout + j0] = acc;
            }
        }
    }
}

// SnakeBeta (snake_beta.rs:37-73): x + sin^2(x * exp(alpha)) / (exp(beta) + 1e-9)
static void snake_beta(std::vector<float>& x, int C, int T, const Tensor& alpha, const Tensor& beta) {
#pragma omp parallel for schedule(static)
    for (int c = 0; c < C; ++c) {
        const float a = std::exp(alpha.v[(size_t)c]);
        const float bden = std::exp(beta.v[(size_t)c]) + (float)1e-9;
        for (int t = 0; t < T; ++t) {
            float v = x[(size_t)c * T + t];
            float sn = std::sin(v * a);
            x[(size_t)c * T + t] = v + (sn * sn) / bden;
        }
    }
}

static inline float gelu_tanh(float v) {  // candle Tensor::gelu (convnext.rs:81)
    return 0.5f * v * (1.0f + std::tanh(0.7978845608028654f * v * (1.0f + 0.044715f * v * v)));
}

void CodecModel::finalize() {
    const auto& c = cfg;
    head_dim = c.head_dim > 0 ? c.head_dim : c.hidden / c.heads;
    // EuclideanCodebook::embeddings (quantizer.rs:56-59)
    codebooks.clear();
    for (int q = 0; q < c.num_quantizers; ++q) {
        std::string p = q == 0 ? "decoder.quantizer.rvq_first.vq.layers.0._codebook."
                               : "decoder.quantizer.rvq_rest.vq.layers." + std::to_string(q - 1) + "._codebook.";
        const Tensor& usage = W(w, p + "cluster_usage");
        const Tensor& sum = W(w, p + "embedding_sum");
        const int dim = (int)sum.dim(1);
        std::vector<float> e(sum.v.size());
        for (int64_t r = 0; r < sum.dim(0); ++r) {
            float u = std::max(usage.v[(size_t)r], 1e-5f);
            for (int i = 0; i < dim; ++i) e[(size_t)r * dim + i] = sum.v[(size_t)r * dim + i] / u;
        }
        codebooks.push_back(std::move(e));
    }
}

void CodecModel::transformer(std::vector<float>& x /* [T][latent] */, int T) const {
    const auto& c = cfg;
    const int Hd = c.hidden, nh = c.heads, nkv = c.kv_heads, d = head_dim, half = d / 2, I = c.inter;
    const std::string P = "decoder.pre_transformer.";
    std::vector<float> h((size_t)T * Hd);
    linear(x.data(), W(w, P + "input_proj.weight").p(), W(w, P + "input_proj.bias").p(), h.data(), T, Hd, c.latent);
    // SimpleRotaryEmbedding (rope/simple.rs:30-64): inv_freq = 1 / theta_f32^(i/half) in f32
    std::vector<float> cs((size_t)T * half), sn((size_t)T * half);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < half; ++i) {
            float inv = 1.0f / std::pow((float)c.rope_theta, (float)i / (float)half);
            float f = (float)t * inv;
            cs[(size_t)t * half + i] = std::cos(f);
            sn[(size_t)t * half + i] = std::sin(f);
        }
    const float scale = (float)std::pow((double)d, -0.5);
    const int rep = nh / nkv;
    std::vector<float> xn((size_t)T * Hd), q((size_t)T * nh * d), k((size_t)T * nkv * d), v((size_t)T * nkv * d),
        attn((size_t)T * nh * d), o((size_t)T * Hd), g((size_t)T * I), u((size_t)T * I);
    for (int l = 0; l < c.layers; ++l) {
        const std::string L = P + "layers." + std::to_string(l) + ".";
        rmsnorm(h.data(), W(w, L + "input_layernorm.weight").p(), xn.data(), T, Hd, c.eps);
        auto bias = [&](const char* n) -> const float* {
            std::string key = L + "self_attn." + n + ".bias";
            return (c.attention_bias && has(w, key)) ? W(w, key).p() : nullptr;
        };
        linear(xn.data(), W(w, L + "self_attn.q_proj.weight").p(), bias("q_proj"), q.data(), T, nh * d, Hd);
        linear(xn.data(), W(w, L + "self_attn.k_proj.weight").p(), bias("k_proj"), k.data(), T, nkv * d, Hd);
        linear(xn.data(), W(w, L + "self_attn.v_proj.weight").p(), bias("v_proj"), v.data(), T, nkv * d, Hd);
        // apply_rotary_pos_emb_manual (rope/mod.rs:78-105): x*cos + rotate_half(x)*sin with cat(c,c)
        auto rot = [&](float* hh, int t) {
            for (int i = 0; i < half; ++i) {
                float x1 = hh[i], x2 = hh[i + half];
                float cc = cs[(size_t)t * half + i], ss = sn[(size_t)t * half + i];
                hh[i] = x1 * cc + (-x2) * ss;
                hh[i + half] = x2 * cc + x1 * ss;
            }
        };
        for (int t = 0; t < T; ++t) {
            for (int hh = 0; hh < nh; ++hh) rot(&q[((size_t)t * nh + hh) * d], t);
            for (int hh = 0; hh < nkv; ++hh) rot(&k[((size_t)t * nkv + hh) * d], t);
        }
        // mask = causal and 0 <= i-j < window (attention/mod.rs:238-266 via transformer.rs:371-377)
#pragma omp parallel
        {
            std::vector<float> sc((size_t)T);
#pragma omp for schedule(static)
            for (int t = 0; t < T; ++t)
                for (int hh = 0; hh < nh; ++hh) {
                    const int gk = hh / rep;
                    const float* qh = &q[((size_t)t * nh + hh) * d];
                    for (int j = 0; j < T; ++j) {
                        float sv = dot(qh, &k[((size_t)j * nkv + gk) * d], d) * scale + 0.0f;
                        int dist = t - j;
                        if (!(dist >= 0 && dist < c.window)) sv += -INFINITY;
                        sc[(size_t)j] = sv;
                    }
                    softmax_inplace(sc.data(), T);
                    float* oo = &attn[((size_t)t * nh + hh) * d];
                    for (int i = 0; i < d; ++i) oo[i] = 0.f;
                    for (int j = 0; j < T; ++j) {
                        const float pj = sc[(size_t)j];
                        const float* vr = &v[((size_t)j * nkv + gk) * d];
                        for (int i = 0; i < d; ++i) oo[i] += pj * vr[i];
                    }
                }
        }
        linear(attn.data(), W(w, L + "self_attn.o_proj.weight").p(), bias("o_proj"), o.data(), T, Hd, nh * d);
        const float* ls = W(w, L + "self_attn_layer_scale.scale").p();
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Hd; ++i) h[(size_t)t * Hd + i] = h[(size_t)t * Hd + i] + o[(size_t)t * Hd + i] * ls[i];
        rmsnorm(h.data(), W(w, L + "post_attention_layernorm.weight").p(), xn.data(), T, Hd, c.eps);
        linear(xn.data(), W(w, L + "mlp.gate_proj.weight").p(), nullptr, g.data(), T, I, Hd);
        linear(xn.data(), W(w, L + "mlp.up_proj.weight").p(), nullptr, u.data(), T, I, Hd);
        for (size_t i = 0; i < (size_t)T * I; ++i) g[i] = silu(g[i]) * u[i];
        linear(g.data(), W(w, L + "mlp.down_proj.weight").p(), nullptr, o.data(), T, Hd, I);
        const float* ms = W(w, L + "mlp_layer_scale.scale").p();
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Hd; ++i) h[(size_t)t * Hd + i] = h[(size_t)t * Hd + i] + o[(size_t)t * Hd + i] * ms[i];
    }
    rmsnorm(h.data(), W(w, P + "norm.weight").p(), xn.data(), T, Hd, c.eps);
    x.assign((size_t)T * c.latent, 0.f);
    linear(xn.data(), W(w, P + "output_proj.weight").p(), W(w, P + "output_proj.bias").p(), x.data(), T, c.latent, Hd);
}

void CodecModel::convnext(std::vector<float>& x, int C, int T, const std::string& p) const {
    std::vector<float> y;
    causal_conv1d(x, C, T, W(w, p + "dwconv.conv.weight"), W(w, p + "dwconv.conv.bias").p(), 1, C, y);
    // LayerNorm over channels (eps 1e-6, affine), time-major rows
    std::vector<float> rows((size_t)T * C), h1((size_t)T * 4 * C), h2((size_t)T * C);
    const float* lw = W(w, p + "norm.weight").p();
    const float* lb = W(w, p + "norm.bias").p();
    for (int t = 0; t < T; ++t) {
        float mean = 0.f;
        for (int c = 0; c < C; ++c) mean += y[(size_t)c * T + t];
        mean /= (float)C;
        float var = 0.f;
        for (int c = 0; c < C; ++c) {
            float dlt = y[(size_t)c * T + t] - mean;
            var += dlt * dlt;
        }
        var /= (float)C;
        float den = std::sqrt(var + 1e-6f);
        for (int c = 0; c < C; ++c) rows[(size_t)t * C + c] = (y[(size_t)c * T + t] - mean) / den * lw[c] + lb[c];
    }
    probe_round(rows.data(), rows.size(), 12);
    linear(rows.data(), W(w, p + "pwconv1.weight").p(), W(w, p + "pwconv1.bias").p(), h1.data(), T, 4 * C, C);
    for (auto& v : h1) v = gelu_tanh(v);
    probe_round(h1.data(), h1.size(), 13);
    linear(h1.data(), W(w, p + "pwconv2.weight").p(), W(w, p + "pwconv2.bias").p(), h2.data(), T, C, 4 * C);
    const float* gm = W(w, p + "gamma").p();
    for (int c = 0; c < C; ++c)
        for (int t = 0; t < T; ++t) x[(size_t)c * T + t] = x[(size_t)c * T + t] + h2[(size_t)t * C + c] * gm[c];
}

// TokenizerV2Decoder::forward (decoder/v2.rs:163-247) for one utterance; codes are [T][Q].
void CodecModel::decode(const uint32_t* codes, int T, std::vector<float>& wav, CodecTrace* trace) const {
    const auto& c = cfg;
    const int Q = c.num_quantizers, cd2 = c.codebook_dim / 2, Dc = c.codebook_dim;
    // SplitResidualVectorQuantizer::decode (quantizer.rs:417-434): sum lookups per group, then each
    // group's 1x1 output_proj, then add the two groups.
    std::vector<float> s1((size_t)T * cd2, 0.f), s2((size_t)T * cd2, 0.f), o1((size_t)T * Dc), o2((size_t)T * Dc);
    for (int t = 0; t < T; ++t) {
        for (int q = 0; q < Q; ++q) {
            uint32_t id = codes[(size_t)t * Q + q];
            if ((int)id >= c.codebook_size) throw std::runtime_error("oracle: code id out of range");
            const float* e = &codebooks[(size_t)q][(size_t)id * cd2];
            float* dst = q == 0 ? &s1[(size_t)t * cd2] : &s2[(size_t)t * cd2];
            if (q <= 1) for (int i = 0; i < cd2; ++i) dst[i] = e[i];
            else for (int i = 0; i < cd2; ++i) dst[i] = dst[i] + e[i];
        }
    }
    const Tensor& p1 = W(w, "decoder.quantizer.rvq_first.output_proj.weight");
    const Tensor& p2 = W(w, "decoder.quantizer.rvq_rest.output_proj.weight");
    linear(s1.data(), p1.p(), nullptr, o1.data(), T, Dc, cd2);
    std::vector<float> x((size_t)Dc * T);
    if (Q > 1) {
        linear(s2.data(), p2.p(), nullptr, o2.data(), T, Dc, cd2);
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Dc; ++i) x[(size_t)i * T + t] = o1[(size_t)t * Dc + i] + o2[(size_t)t * Dc + i];
    } else {
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Dc; ++i) x[(size_t)i * T + t] = o1[(size_t)t * Dc + i];
    }
    if (trace) trace->after_quantizer = x;
    probe_round(x.data(), x.size(), 8);
    std::vector<float> y;
    causal_conv1d(x, Dc, T, W(w, "decoder.pre_conv.conv.weight"), W(w, "decoder.pre_conv.conv.bias").p(), 1, 1, y);
    if (trace) trace->after_pre_conv = y;
    probe_round(y.data(), y.size(), 9);
    // transformer on [T][latent]
    std::vector<float> seq((size_t)T * c.latent);
    for (int ch = 0; ch < c.latent; ++ch)
        for (int t = 0; t < T; ++t) seq[(size_t)t * c.latent + ch] = y[(size_t)ch * T + t];
    transformer(seq, T);
    x.assign((size_t)c.latent * T, 0.f);
    for (int ch = 0; ch < c.latent; ++ch)
        for (int t = 0; t < T; ++t) x[(size_t)ch * T + t] = seq[(size_t)t * c.latent + ch];
    if (trace) trace->after_transformer = x;
    probe_round(x.data(), x.size(), 10);
    int Tc = T, C = c.latent;
    for (size_t i = 0; i < c.upsampling_ratios.size(); ++i) {
        const std::string p = "decoder.upsample." + std::to_string(i) + ".";
        int To;
        causal_conv_transpose1d(x, C, Tc, W(w, p + "0.conv.weight"), W(w, p + "0.conv.bias").p(), c.upsampling_ratios[i], y, To);
        x.swap(y);
        Tc = To;
        probe_round(x.data(), x.size(), 11);
        convnext(x, C, Tc, p + "1.");
        probe_round(x.data(), x.size(), 14);
    }
    if (trace) trace->after_upsample = x;
    causal_conv1d(x, C, Tc, W(w, "decoder.decoder.0.conv.weight"), W(w, "decoder.decoder.0.conv.bias").p(), 1, 1, y);
    x.swap(y);
    C = c.decoder_dim;
    for (size_t b = 0; b < c.upsample_rates.size(); ++b) {  // DecoderBlock (residual_unit.rs:47-96)
        const std::string p = "decoder.decoder." + std::to_string(b + 1) + ".block.";
        snake_beta(x, C, Tc, W(w, p + "0.alpha"), W(w, p + "0.beta"));
        probe_round(x.data(), x.size(), 15);
        probe_round(x.data(), x.size(), 19 + (int)b);
        int To;
        causal_conv_transpose1d(x, C, Tc, W(w, p + "1.conv.weight"), W(w, p + "1.conv.bias").p(), c.upsample_rates[b], y, To);
        x.swap(y);
        Tc = To;
        C /= 2;
        const int dils[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {  // DecoderResidualUnit (residual_unit.rs:11-43)
            const std::string q = p + std::to_string(u + 2) + ".";
            std::vector<float> hcur(x);
            snake_beta(hcur, C, Tc, W(w, q + "act1.alpha"), W(w, q + "act1.beta"));
            probe_round(hcur.data(), hcur.size(), 16);
            causal_conv1d(hcur, C, Tc, W(w, q + "conv1.conv.weight"), W(w, q + "conv1.conv.bias").p(), dils[u], 1, y);
            snake_beta(y, C, Tc, W(w, q + "act2.alpha"), W(w, q + "act2.beta"));
            probe_round(y.data(), y.size(), 17);
            causal_conv1d(y, C, Tc, W(w, q + "conv2.conv.weight"), W(w, q + "conv2.conv.bias").p(), 1, 1, hcur);
            for (size_t i = 0; i < x.size(); ++i) x[i] = x[i] + hcur[i];
        }
        if (trace) trace->after_block.push_back(x);
    }
    const int nb = (int)c.upsample_rates.size();
    snake_beta(x, C, Tc, W(w, "decoder.decoder." + std::to_string(nb + 1) + ".alpha"),
               W(w, "decoder.decoder." + std::to_string(nb + 1) + ".beta"));
    probe_round(x.data(), x.size(), 18);
    causal_conv1d(x, C, Tc, W(w, "decoder.decoder." + std::to_string(nb + 2) + ".conv.weight"),
                  W(w, "decoder.decoder." + std::to_string(nb + 2) + ".conv.bias").p(), 1, 1, y);
    wav.resize((size_t)Tc);
    for (int t = 0; t < Tc; ++t) wav[(size_t)t] = std::min(1.0f, std::max(-1.0f, y[(size_t)t]));
}

// chunked_decode (decoder/v2.rs:250-286)
void CodecModel::chunked_decode(const uint32_t* codes, int T, int chunk, int left_ctx, std::vector<float>& wav) const {
    wav.clear();
    int total_up = 1;
    for (int r : cfg.upsample_rates) total_up *= r;
    for (int r : cfg.upsampling_ratios) total_up *= r;
    int start = 0;
    while (start < T) {
        int end = std::min(start + chunk, T);
        int ctx = start > left_ctx ? left_ctx : start;
        std::vector<float> wchunk;
        decode(codes + (size_t)(start - ctx) * cfg.num_quantizers, end - start + ctx, wchunk, nullptr);
        size_t drop = (size_t)ctx * total_up;
        if (drop > wchunk.size()) throw std::runtime_error("oracle: chunk shorter than its context");
        wav.insert(wav.end(), wchunk.begin() + (long)drop, wchunk.end());
        start = end;
    }
}

// Model::decode_output policy at batch 1 (model/generate.rs:71-111)
void CodecModel::decode_output(const uint32_t* codes, int len, std::vector<float>& wav) const {
    if (len > 512) chunked_decode(codes, len, 256, 32, wav);
    else decode(codes, len, wav, nullptr);
}

}  // namespace orc
