// CPU oracle (test infrastructure): 12.5 Hz codec ENCODER (waveform -> codes).
// Restates qwen_tts/src/audio/encoder/{v2,seanet,transformer,quantizer}.rs.  Activations are channels-first
// [C][T] per utterance, as in the reference.  The reference holds no numeric vectors for this path; the restatement is pinned
// instead on codes computed by an independent implementation of the same architecture (transformers' MimiModel on the same
// synthetic weights, tests/golden/make_mimi_golden.py), adjusted for the two places where the reference departs from Mimi:
// the stride-2 downsample conv is zero-padded (seanet.rs:244-248; Mimi replicates) and a partial last frame is dropped.
#include "oracle_codec.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace orc {

namespace {

// conv1d with left zero padding `pad`, stride, dilation (candle conv1d, weight (Cout, Cin, k)); bias optional
void conv1d_causal(const std::vector<float>& x, int Cin, int T, const Tensor& w, const float* b, int stride, int dil, int pad,
                   std::vector<float>& y, int& Tout) {
    const int Cout = (int)w.dim(0), k = (int)w.dim(2);
    if ((int)w.dim(1) != Cin) throw std::runtime_error("oracle: encoder conv channel mismatch");
    const int eff = (k - 1) * dil + 1;
    Tout = (T + pad - eff) / stride + 1;
    if (T + pad < eff) Tout = 0;
    y.assign((size_t)Cout * std::max(Tout, 0), 0.f);
#pragma omp parallel
    {
        std::vector<float> patch((size_t)Cin * k);
#pragma omp for schedule(static)
        for (int t = 0; t < Tout; ++t) {
            for (int ci = 0; ci < Cin; ++ci)
                for (int kk = 0; kk < k; ++kk) {
                    const int xi = t * stride + kk * dil - pad;
                    patch[(size_t)ci * k + kk] = xi >= 0 ? x[(size_t)ci * T + xi] : 0.f;
                }
            for (int co = 0; co < Cout; ++co) {
                float s = dot(patch.data(), w.p() + (size_t)co * Cin * k, (int64_t)Cin * k);
                if (b) s += b[co];
                y[(size_t)co * Tout + t] = s;
            }
        }
    }
}

inline float elu(float v) { return v > 0.f ? v : std::exp(v) - 1.0f; }
inline float gelu_erf(float v) { return 0.5f * v * (1.0f + std::erf(v * 0.70710678118654752440f)); }

void layer_norm(const float* x, const float* w, const float* b, float* y, int T, int D, float eps) {
    for (int t = 0; t < T; ++t) {
        const float* r = x + (size_t)t * D;
        float mean = 0.f;
        for (int i = 0; i < D; ++i) mean += r[i];
        mean /= (float)D;
        float var = 0.f;
        for (int i = 0; i < D; ++i) {
            const float d = r[i] - mean;
            var += d * d;
        }
        var /= (float)D;
        const float den = std::sqrt(var + eps);
        for (int i = 0; i < D; ++i) y[(size_t)t * D + i] = (r[i] - mean) / den * w[i] + b[i];
    }
}

}  // namespace

// TokenizerV2Encoder::encode (audio/encoder/v2.rs:87-120) for one utterance; codes come out [T][valid_q].
// margins (optional): per (frame, codebook) the distance gap between the nearest and the second nearest entry, and the
// magnitude |e|^2 / 2 + |x.e| of the winning distance's terms - what a parity test needs to tell a rounding-level near-tie from a
// wrong code.
void encode_audio(const Weights& w, const float* audio, int n, int valid_q, std::vector<uint32_t>& codes, int& T_out,
                  std::vector<float>* margins) {
    const std::string E = "encoder.encoder.layers.";
    // ---- SeaNetEncoder::forward (seanet.rs:153-176)
    std::vector<float> x(audio, audio + n), y;
    int T = n, To = 0, C = 64;
    conv1d_causal(x, 1, T, W(w, E + "0.conv.weight"), W(w, E + "0.conv.bias").p(), 1, 1, 6, y, To);
    x.swap(y);
    T = To;
    const int ratios[4] = {4, 5, 6, 8};
    int idx = 1;
    for (int s = 0; s < 4; ++s) {
        // SeaNetResBlock (seanet.rs:62-77): x + conv2(elu(conv1(pad(elu(x)))))
        std::vector<float> a(x);
        for (auto& v : a) v = elu(v);
        std::vector<float> h1, h2;
        int t1, t2;
        const std::string R = E + std::to_string(idx) + ".block.";
        conv1d_causal(a, C, T, W(w, R + "1.conv.weight"), W(w, R + "1.conv.bias").p(), 1, 1, 2, h1, t1);
        for (auto& v : h1) v = elu(v);
        conv1d_causal(h1, C / 2, t1, W(w, R + "3.conv.weight"), W(w, R + "3.conv.bias").p(), 1, 1, 0, h2, t2);
        for (size_t i = 0; i < x.size(); ++i) x[i] = x[i] + h2[i];
        // ELU + strided downsample conv (kernel 2r, stride r, left pad r)
        for (auto& v : x) v = elu(v);
        const std::string D = E + std::to_string(idx + 2) + ".conv.";
        conv1d_causal(x, C, T, W(w, D + "weight"), W(w, D + "bias").p(), ratios[s], 1, ratios[s], y, To);
        x.swap(y);
        T = To;
        C *= 2;
        idx += 3;
    }
    for (auto& v : x) v = elu(v);
    conv1d_causal(x, C, T, W(w, E + std::to_string(idx + 1) + ".conv.weight"), W(w, E + std::to_string(idx + 1) + ".conv.bias").p(), 1, 1, 2, y, To);
    x.swap(y);
    T = To;
    const int Dm = 512, nh = 8, d = 64, half = 32, I = 2048;
    if (T < 2) throw std::runtime_error("oracle: audio too short for the encoder");
    // ---- EncoderTransformer (transformer.rs:170-184, seanet.rs:207-213) on [T][512]
    std::vector<float> h((size_t)T * Dm);
    for (int c = 0; c < Dm; ++c)
        for (int t = 0; t < T; ++t) h[(size_t)t * Dm + c] = x[(size_t)c * T + t];
    std::vector<float> cs((size_t)T * half), sn((size_t)T * half);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < half; ++i) {  // SimpleRotaryEmbedding (rope/simple.rs:30-64), theta 10000
            const float inv = 1.0f / std::pow(10000.0f, (float)i / (float)half);
            const float f = (float)t * inv;
            cs[(size_t)t * half + i] = std::cos(f);
            sn[(size_t)t * half + i] = std::sin(f);
        }
    std::vector<float> xn((size_t)T * Dm), q((size_t)T * Dm), k((size_t)T * Dm), v((size_t)T * Dm), at((size_t)T * Dm), o((size_t)T * Dm),
        f1((size_t)T * I);
    const float sq = (float)std::sqrt((double)d);
    for (int l = 0; l < 8; ++l) {
        const std::string L = "encoder.encoder_transformer.layers." + std::to_string(l) + ".";
        layer_norm(h.data(), W(w, L + "input_layernorm.weight").p(), W(w, L + "input_layernorm.bias").p(), xn.data(), T, Dm, 1e-5f);
        linear(xn.data(), W(w, L + "self_attn.q_proj.weight").p(), nullptr, q.data(), T, Dm, Dm);
        linear(xn.data(), W(w, L + "self_attn.k_proj.weight").p(), nullptr, k.data(), T, Dm, Dm);
        linear(xn.data(), W(w, L + "self_attn.v_proj.weight").p(), nullptr, v.data(), T, Dm, Dm);
        auto rot = [&](float* hh, int t) {  // apply_rotary_pos_emb_manual (rope/mod.rs:78-105)
            for (int i = 0; i < half; ++i) {
                const float x1 = hh[i], x2 = hh[i + half];
                const float cc = cs[(size_t)t * half + i], ss = sn[(size_t)t * half + i];
                hh[i] = x1 * cc + (-x2) * ss;
                hh[i + half] = x2 * cc + x1 * ss;
            }
        };
        for (int t = 0; t < T; ++t)
            for (int hh = 0; hh < nh; ++hh) {
                rot(&q[((size_t)t * nh + hh) * d], t);
                rot(&k[((size_t)t * nh + hh) * d], t);
            }
#pragma omp parallel
        {
            std::vector<float> sc((size_t)T);
#pragma omp for schedule(static)
            for (int t = 0; t < T; ++t)
                for (int hh = 0; hh < nh; ++hh) {
                    const float* qh = &q[((size_t)t * nh + hh) * d];
                    for (int j = 0; j < T; ++j) {
                        float sv = dot(qh, &k[((size_t)j * nh + hh) * d], d) / sq;  // (q k^T) / sqrt(d), then the causal mask
                        if (j > t) sv += -INFINITY;
                        sc[(size_t)j] = sv;
                    }
                    softmax_inplace(sc.data(), T);
                    float* oo = &at[((size_t)t * nh + hh) * d];
                    for (int i = 0; i < d; ++i) oo[i] = 0.f;
                    for (int j = 0; j < T; ++j) {
                        const float pj = sc[(size_t)j];
                        const float* vr = &v[((size_t)j * nh + hh) * d];
                        for (int i = 0; i < d; ++i) oo[i] += pj * vr[i];
                    }
                }
        }
        linear(at.data(), W(w, L + "self_attn.o_proj.weight").p(), nullptr, o.data(), T, Dm, Dm);
        const float* ls = W(w, L + "self_attn_layer_scale.scale").p();
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Dm; ++i) h[(size_t)t * Dm + i] = h[(size_t)t * Dm + i] + o[(size_t)t * Dm + i] * ls[i];
        layer_norm(h.data(), W(w, L + "post_attention_layernorm.weight").p(), W(w, L + "post_attention_layernorm.bias").p(), xn.data(), T, Dm, 1e-5f);
        linear(xn.data(), W(w, L + "mlp.fc1.weight").p(), nullptr, f1.data(), T, I, Dm);
        for (auto& e : f1) e = gelu_erf(e);
        linear(f1.data(), W(w, L + "mlp.fc2.weight").p(), nullptr, o.data(), T, Dm, I);
        const float* ms = W(w, L + "mlp_layer_scale.scale").p();
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < Dm; ++i) h[(size_t)t * Dm + i] = h[(size_t)t * Dm + i] + o[(size_t)t * Dm + i] * ms[i];
    }
    // ---- Downsample (seanet.rs:243-249): conv k=4, stride 2, no bias, left pad 2
    x.assign((size_t)Dm * T, 0.f);
    for (int c = 0; c < Dm; ++c)
        for (int t = 0; t < T; ++t) x[(size_t)c * T + t] = h[(size_t)t * Dm + c];
    conv1d_causal(x, Dm, T, W(w, "encoder.downsample.conv.weight"), nullptr, 2, 1, 2, y, To);
    T = To;
    if (T < 1) throw std::runtime_error("oracle: audio too short for the encoder");
    // ---- EncoderSplitRVQ::encode (quantizer.rs:117-179): nearest neighbour by argmin(|e|^2 / 2 - x.e), residual update
    std::vector<float> rows((size_t)T * Dm);
    for (int c = 0; c < Dm; ++c)
        for (int t = 0; t < T; ++t) rows[(size_t)t * Dm + c] = y[(size_t)c * T + t];
    codes.assign((size_t)T * valid_q, 0);
    if (margins) margins->assign((size_t)T * valid_q * 2, 0.f);
    int qi = 0;
    for (int grp = 0; grp < 2 && qi < valid_q; ++grp) {
        const std::string P = grp == 0 ? "encoder.quantizer.semantic_residual_vector_quantizer." : "encoder.quantizer.acoustic_residual_vector_quantizer.";
        const int nlayers = grp == 0 ? 1 : 31;
        const Tensor& ip = W(w, P + "input_proj.weight");
        const int cd = (int)ip.dim(0);
        std::vector<float> res((size_t)T * cd);
        linear(rows.data(), ip.p(), nullptr, res.data(), T, cd, Dm);
        for (int l = 0; l < nlayers && qi < valid_q; ++l, ++qi) {
            const std::string Cb = P + "layers." + std::to_string(l) + ".codebook.";
            const Tensor& sum = W(w, Cb + "embed_sum");
            const Tensor& usage = W(w, Cb + "cluster_usage");
            const int bins = (int)sum.dim(0);
            std::vector<float> emb(sum.v.size()), c2((size_t)bins);
            for (int r = 0; r < bins; ++r) {
                const float u = std::max(usage.v[(size_t)r], 1e-5f);
                float s2 = 0.f;
                for (int i = 0; i < cd; ++i) {
                    const float e = sum.v[(size_t)r * cd + i] / u;
                    emb[(size_t)r * cd + i] = e;
                    s2 += e * e;
                }
                c2[(size_t)r] = s2 * 0.5f + 0.0f;
            }
#pragma omp parallel for schedule(static)
            for (int t = 0; t < T; ++t) {
                float best = INFINITY, second = INFINITY, best_dot = 0.f;
                int bi = 0;
                for (int r = 0; r < bins; ++r) {
                    const float d = dot(&res[(size_t)t * cd], &emb[(size_t)r * cd], cd);
                    const float dist = c2[(size_t)r] - d;
                    if (dist < best) { second = best; best = dist; bi = r; best_dot = d; }  // argmin keeps the first minimum
                    else if (dist < second) second = dist;
                }
                codes[(size_t)t * valid_q + qi] = (uint32_t)bi;
                if (margins) {
                    (*margins)[((size_t)t * valid_q + qi) * 2] = second - best;
                    (*margins)[((size_t)t * valid_q + qi) * 2 + 1] = c2[(size_t)bi] + std::fabs(best_dot);
                }
                for (int i = 0; i < cd; ++i) res[(size_t)t * cd + i] = res[(size_t)t * cd + i] - emb[(size_t)bi * cd + i];
            }
        }
    }
    T_out = T;
}

}  // namespace orc
