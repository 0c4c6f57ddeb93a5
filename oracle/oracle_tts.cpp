// CPU oracle (test infrastructure): talker, code predictor, sampling, RNG, generate loop.
// See oracle.hpp for scope and parity status.
#include "oracle.hpp"

#include <algorithm>
#include <cassert>
#include <stdexcept>

namespace orc {

const Tensor& W(const Weights& w, const std::string& key) {
    auto it = w.find(key);
    if (it == w.end()) throw std::runtime_error("oracle: missing tensor " + key);
    return it->second;
}
bool has(const Weights& w, const std::string& key) { return w.find(key) != w.end(); }

void linear(const float* x, const float* w, const float* b, float* y, int64_t M, int64_t N, int64_t K) {
#pragma omp parallel for schedule(static)
    for (int64_t n = 0; n < N; ++n) {
        const float* wr = w + n * K;
        for (int64_t m = 0; m < M; ++m) {
            float s = dot(x + m * K, wr, K);
            if (b) s += b[n];
            y[m * N + n] = s;
        }
    }
}

void rmsnorm(const float* x, const float* w, float* y, int64_t rows, int64_t n, float eps) {
    for (int64_t r = 0; r < rows; ++r) {
        const float* xr = x + r * n;
        float* yr = y + r * n;
        float sum2 = 0.f;
        for (int64_t i = 0; i < n; ++i) sum2 += xr[i] * xr[i];
        float m = std::sqrt(sum2 / (float)n + eps);
        for (int64_t i = 0; i < n; ++i) yr[i] = xr[i] / m * w[i];
    }
}

void softmax_inplace(float* x, int64_t n) {
    float mx = -INFINITY;
    for (int64_t i = 0; i < n; ++i) mx = std::max(mx, x[i]);
    float sum = 0.f;
    for (int64_t i = 0; i < n; ++i) {
        x[i] = std::exp(x[i] - mx);
        sum += x[i];
    }
    for (int64_t i = 0; i < n; ++i) x[i] /= sum;
}

// ------------------------------------------------------------------------------------ RNG
void MT19937::reseed(uint32_t seed) {
    mt[0] = seed;
    for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    idx = 624;
}
uint32_t MT19937::next_u32() {
    if (idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t y = mt[idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void randn_vectorized(MT19937& rng, float* out, int count) {  // mt_rng.rs:116-148
    for (int c = 0; c + 16 <= count; c += 16) {
        float u[16];
        for (int i = 0; i < 16; ++i) u[i] = rng.uniform();
        for (int i = 0; i < 8; ++i) {
            float u1 = 1.0f - u[i];
            float u2 = u[8 + i];
            float r = std::sqrt(-2.0f * std::log(u1));
            float th = 2.0f * 3.14159265358979323846f * u2;
            out[c + i] = r * std::cos(th);
            out[c + 8 + i] = r * std::sin(th);
        }
    }
}

// ------------------------------------------------------------------------------- sampling
SamplingConfig SamplingConfig::for_subtalker() const {  // sampling.rs:109-125
    SamplingConfig s = *this;
    s.temperature = subtalker_temperature;
    s.top_k = subtalker_top_k;
    s.top_p = subtalker_top_p;
    s.do_sample = subtalker_do_sample;
    s.repetition_penalty = 1.0;
    s.suppress_lo = s.suppress_hi = 0;
    s.min_new_tokens = 0;
    s.eos_token_id = -1;
    return s;
}

void apply_repetition_penalty(std::vector<float>& l, const std::vector<int64_t>& hist, double penalty) {
    if (penalty == 1.0 || hist.empty()) return;  // sampling.rs:178-180
    const float p = (float)penalty;
    for (int64_t t : hist) {  // every occurrence is penalised again (sampling.rs:184-193)
        if (t >= 0 && (size_t)t < l.size()) {
            if (l[(size_t)t] > 0.0f) l[(size_t)t] /= p;
            else l[(size_t)t] *= p;
        }
    }
}
void apply_temperature(std::vector<float>& l, double t) {
    if (t == 1.0) return;
    const float m = (float)(1.0 / t);  // Tensor::affine(1/t, 0): mul cast to the tensor dtype
    for (auto& v : l) v = v * m + 0.0f;
}
void apply_top_k(std::vector<float>& l, int k) {  // sampling.rs:209-229
    if (k == 0 || (size_t)k >= l.size()) return;
    std::vector<float> s = l;
    std::sort(s.begin(), s.end(), [](float a, float b) { return a > b; });
    float thr = s[(size_t)k - 1];
    for (auto& v : l)
        if (v < thr) v = -INFINITY;
}
void apply_top_p(std::vector<float>& l, double p) {  // sampling.rs:234-274
    if (p >= 1.0) return;
    const size_t n = l.size();
    float mx = -INFINITY;
    for (float v : l) mx = std::max(mx, v);
    std::vector<float> e(n);
    float sum = 0.f;
    for (size_t i = 0; i < n; ++i) {
        e[i] = std::exp(l[i] - mx);
        sum += e[i];
    }
    for (size_t i = 0; i < n; ++i) e[i] /= sum;
    std::vector<size_t> idx(n);
    for (size_t i = 0; i < n; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return e[a] > e[b]; });
    float cum = 0.f;
    size_t cutoff = n;
    for (size_t i = 0; i < n; ++i) {
        cum += e[idx[i]];
        if (cum > (float)p) {
            cutoff = i + 1;
            break;
        }
    }
    std::vector<float> f(n, -INFINITY);
    for (size_t i = 0; i < cutoff; ++i) f[idx[i]] = l[idx[i]];
    l.swap(f);
}

// Iterator::max_by keeps the LAST of several equal maxima (fallbacks at sampling.rs:370-375, 391-396)
static int64_t argmax_last(const std::vector<float>& v) {
    int64_t best = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] >= v[(size_t)best]) best = (int64_t)i;
    return best;
}

static int64_t argmax_first(const std::vector<float>& v) {
    int64_t best = 0;
    for (size_t i = 1; i < v.size(); ++i)
        if (v[i] > v[(size_t)best]) best = (int64_t)i;
    return best;
}

int64_t sample_token(const float* logits, int64_t V, const SamplingConfig& cfg,
                     const std::vector<int64_t>& history, MT19937* rng) {
    std::vector<float> l(logits, logits + V);
    apply_repetition_penalty(l, history, cfg.repetition_penalty);
    for (int i = cfg.suppress_lo; i < cfg.suppress_hi && i < V; ++i)
        if (i != cfg.eos_token_id) l[(size_t)i] = -INFINITY;
    if ((int)history.size() < cfg.min_new_tokens && cfg.eos_token_id >= 0 && cfg.eos_token_id < V)
        l[(size_t)cfg.eos_token_id] = -INFINITY;  // sampling.rs:319-326
    apply_temperature(l, cfg.temperature);
    apply_top_k(l, cfg.top_k);
    apply_top_p(l, cfg.top_p);
    if (!cfg.do_sample) return argmax_first(l);
    softmax_inplace(l.data(), V);  // candle_nn::ops::softmax(&logits, 0)
    // sample_from_probs (sampling.rs:362-397)
    float sum = 0.f;
    for (float p : l) sum += p;
    if (!(sum > 0.0f) || !std::isfinite(sum)) return argmax_last(l);
    float u = rng->uniform();
    float cum = 0.f, inv = 1.0f / sum;
    for (int64_t i = 0; i < V; ++i) {
        cum += l[(size_t)i] * inv;
        if (u < cum) return i;
    }
    return argmax_last(l);
}

// ----------------------------------------------------------------------------------- rope
void RopeTable::init(int d, double theta) {
    half = d / 2;
    inv_freq.resize((size_t)half);
    for (int i = 0; i < half; ++i) inv_freq[(size_t)i] = 1.0f / (float)std::pow(theta, (double)i * 2.0 / (double)d);
}
void RopeTable::cos_sin(int64_t pos, float* c, float* s) const {
    for (int i = 0; i < half; ++i) {
        float f = inv_freq[(size_t)i] * (float)pos;
        c[i] = std::cos(f);
        s[i] = std::sin(f);
    }
}

// -------------------------------------------------------------------------- decoder layer
// Diagnostic only (0 in every parity check): rounds the named intermediate tensors to bf16 so that the storage precision of
// a reduced-precision implementation can be priced tensor by tensor.  bit 0: normed layer inputs, bit 1: keys / values,
// bit 2: attention output, bit 3: MLP activation; bits 8-18: tensors of the codec decoder (oracle_codec.cpp).
static int g_round_probe = 0;
void set_round_probe(int mask) { g_round_probe = mask; }
void probe_round(float* x, size_t n, int bit) {
    if (!(g_round_probe & (1 << bit))) return;
    for (size_t i = 0; i < n; ++i) {
        uint32_t u;
        std::memcpy(&u, &x[i], 4);
        u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
        std::memcpy(&x[i], &u, 4);
    }
}

void decoder_layer(const Weights& w, const std::string& p, const AttnShape& a, const RopeTable& rope, float* x,
                   int T, int pos0, MaskMode mask, KVCacheLayer& cache) {
    const int H = a.hidden, nh = a.nh, nkv = a.nkv, d = a.d, half = d / 2;
    std::vector<float> xn((size_t)T * H), q((size_t)T * nh * d), k((size_t)T * nkv * d), v((size_t)T * nkv * d);
    rmsnorm(x, W(w, p + "input_layernorm.weight").p(), xn.data(), T, H, a.eps);
    probe_round(xn.data(), xn.size(), 0);
    auto bias = [&](const char* n) -> const float* {
        std::string key = p + "self_attn." + n + ".bias";
        return (a.bias && has(w, key)) ? W(w, key).p() : nullptr;
    };
    linear(xn.data(), W(w, p + "self_attn.q_proj.weight").p(), bias("q_proj"), q.data(), T, nh * d, H);
    linear(xn.data(), W(w, p + "self_attn.k_proj.weight").p(), bias("k_proj"), k.data(), T, nkv * d, H);
    linear(xn.data(), W(w, p + "self_attn.v_proj.weight").p(), bias("v_proj"), v.data(), T, nkv * d, H);
    // per-head RMSNorm over d (unified.rs:199-201), then RoPE (rope/mod.rs:46-62, candle rope:
    // y[i] = x[i] c[i] - x[i+d/2] s[i]; y[i+d/2] = x[i] s[i] + x[i+d/2] c[i])
    rmsnorm(q.data(), W(w, p + "self_attn.q_norm.weight").p(), q.data(), (int64_t)T * nh, d, a.eps);
    rmsnorm(k.data(), W(w, p + "self_attn.k_norm.weight").p(), k.data(), (int64_t)T * nkv, d, a.eps);
    std::vector<float> c((size_t)half), s((size_t)half);
    for (int t = 0; t < T; ++t) {
        rope.cos_sin(pos0 + t, c.data(), s.data());
        auto rot = [&](float* h) {
            for (int i = 0; i < half; ++i) {
                float x1 = h[i], x2 = h[i + half];
                h[i] = x1 * c[(size_t)i] - x2 * s[(size_t)i];
                h[i + half] = x1 * s[(size_t)i] + x2 * c[(size_t)i];
            }
        };
        for (int h = 0; h < nh; ++h) rot(&q[((size_t)t * nh + h) * d]);
        for (int h = 0; h < nkv; ++h) rot(&k[((size_t)t * nkv + h) * d]);
    }
    probe_round(k.data(), k.size(), 1);
    probe_round(v.data(), v.size(), 1);
    // KV append (kv_cache.rs:38-42)
    if (cache.len + T > cache.cap) throw std::runtime_error("oracle: kv cache overflow");
    for (int t = 0; t < T; ++t)
        for (int h = 0; h < nkv; ++h) {
            std::memcpy(&cache.k[((size_t)h * cache.cap + cache.len + t) * d], &k[((size_t)t * nkv + h) * d], sizeof(float) * d);
            std::memcpy(&cache.v[((size_t)h * cache.cap + cache.len + t) * d], &v[((size_t)t * nkv + h) * d], sizeof(float) * d);
        }
    const int kv_len = cache.len + T;
    const int rep = nh / nkv;
    const float scale = (float)std::pow((double)d, -0.5);
    std::vector<float> attn((size_t)T * nh * d), sc((size_t)kv_len);
    for (int t = 0; t < T; ++t) {
        const int qpos = cache.len + t;  // full_q_pos = kv_len - q_len + q_pos
        for (int h = 0; h < nh; ++h) {
            const int g = h / rep;  // repeat_kv: head h uses kv head h / n_rep (attention/mod.rs:168-184)
            const float* qh = &q[((size_t)t * nh + h) * d];
            for (int j = 0; j < kv_len; ++j) {
                float sv = dot(qh, &cache.k[((size_t)g * cache.cap + j) * d], d) * scale + 0.0f;
                if (mask == MASK_CAUSAL && j > qpos) sv += -INFINITY;
                if (a.window > 0) {
                    int dist = qpos - j;
                    if (!(dist >= 0 && dist < a.window)) sv += -INFINITY;
                }
                sc[(size_t)j] = sv;
            }
            softmax_inplace(sc.data(), kv_len);
            float* o = &attn[((size_t)t * nh + h) * d];
            for (int i = 0; i < d; ++i) o[i] = 0.f;
            for (int j = 0; j < kv_len; ++j) {
                const float pj = sc[(size_t)j];
                const float* vr = &cache.v[((size_t)g * cache.cap + j) * d];
                for (int i = 0; i < d; ++i) o[i] += pj * vr[i];
            }
        }
    }
    cache.len = kv_len;
    probe_round(attn.data(), attn.size(), 2);
    std::vector<float> o((size_t)T * H);
    linear(attn.data(), W(w, p + "self_attn.o_proj.weight").p(), bias("o_proj"), o.data(), T, H, nh * d);
    for (size_t i = 0; i < (size_t)T * H; ++i) x[i] = x[i] + o[i];
    // MLP (mlp.rs:59-82): down(silu(gate) * up)
    rmsnorm(x, W(w, p + "post_attention_layernorm.weight").p(), xn.data(), T, H, a.eps);
    probe_round(xn.data(), xn.size(), 0);
    const int I = a.inter;
    std::vector<float> g((size_t)T * I), u((size_t)T * I);
    linear(xn.data(), W(w, p + "mlp.gate_proj.weight").p(), nullptr, g.data(), T, I, H);
    linear(xn.data(), W(w, p + "mlp.up_proj.weight").p(), nullptr, u.data(), T, I, H);
    for (size_t i = 0; i < (size_t)T * I; ++i) g[i] = silu(g[i]) * u[i];
    probe_round(g.data(), g.size(), 3);
    linear(g.data(), W(w, p + "mlp.down_proj.weight").p(), nullptr, o.data(), T, H, I);
    for (size_t i = 0; i < (size_t)T * H; ++i) x[i] = x[i] + o[i];
}

// ---------------------------------------------------------------------------------- model
void Model::finalize() {
    rope_t.init(cfg.talker.d, cfg.talker.theta);
    rope_c.init(cfg.cp.d, cfg.cp.theta);
}

void Model::embed_and_project_text(const uint32_t* ids, int n, float* out) const {
    const int Ht = cfg.text_hidden, H = cfg.talker.hidden;
    const Tensor& emb = W(w, "talker.model.text_embedding.weight");
    std::vector<float> e((size_t)n * Ht), h((size_t)n * Ht);
    for (int i = 0; i < n; ++i) {
        if ((int64_t)ids[i] >= emb.dim(0)) throw std::runtime_error("oracle: text id out of range");
        std::memcpy(&e[(size_t)i * Ht], emb.p() + (size_t)ids[i] * Ht, sizeof(float) * Ht);
    }
    // TalkerResizeMLP (mlp.rs:149-160): fc1(+b) -> silu -> fc2(+b)
    linear(e.data(), W(w, "talker.text_projection.linear_fc1.weight").p(), W(w, "talker.text_projection.linear_fc1.bias").p(),
           h.data(), n, Ht, Ht);
    for (auto& v : h) v = silu(v);
    linear(h.data(), W(w, "talker.text_projection.linear_fc2.weight").p(), W(w, "talker.text_projection.linear_fc2.bias").p(),
           out, n, H, Ht);
}

void Model::sum_code_embeddings(const uint32_t* codes, float* out) const {
    const int H = cfg.talker.hidden;
    if ((int64_t)codes[0] >= W(w, "talker.model.codec_embedding.weight").dim(0)) throw std::runtime_error("oracle: code id out of range");
    const float* e0 = W(w, "talker.model.codec_embedding.weight").p() + (size_t)codes[0] * H;
    for (int i = 0; i < H; ++i) out[i] = e0[i];
    for (int g = 1; g < cfg.groups; ++g) {
        const Tensor& tab = W(w, "talker.code_predictor.model.codec_embedding." + std::to_string(g - 1) + ".weight");
        if ((int64_t)codes[g] >= tab.dim(0)) throw std::runtime_error("oracle: code id out of range");
        const float* e = tab.p() + (size_t)codes[g] * H;
        for (int i = 0; i < H; ++i) out[i] = out[i] + e[i];
    }
}

void Model::build_prompt(const GenRequest& r, std::vector<float>& prompt, int& P, std::vector<float>& trailing, int& Tt,
                         std::vector<float>& tts_pad) const {
    const int H = cfg.talker.hidden;
    const float* cemb = W(w, "talker.model.codec_embedding.weight").p();
    auto codec = [&](int id) { return cemb + (size_t)id * H; };
    auto push = [&](std::vector<float>& dst, const float* a, const float* b) {
        for (int i = 0; i < H; ++i) dst.push_back(b ? a[i] + b[i] : a[i]);
    };
    // 1. tts special embeds (generation.rs:422-442)
    uint32_t sp[3] = {(uint32_t)cfg.tts_bos, (uint32_t)cfg.tts_eos, (uint32_t)cfg.tts_pad};
    std::vector<float> spe((size_t)3 * H);
    embed_and_project_text(sp, 3, spe.data());
    const float* tts_bos = &spe[0];
    const float* tts_eos = &spe[(size_t)H];
    tts_pad.assign(&spe[(size_t)2 * H], &spe[(size_t)2 * H] + H);
    // 2-4. codec input rows: prefill || [speaker] || pad, bos (generation.rs:529-571)
    std::vector<std::vector<float>> ci;
    for (int id : r.codec_prefill) ci.emplace_back(codec(id), codec(id) + H);
    if (!r.speaker_embed.empty()) ci.emplace_back(r.speaker_embed.begin(), r.speaker_embed.begin() + H);
    else if (r.speaker_token >= 0) ci.emplace_back(codec(r.speaker_token), codec(r.speaker_token) + H);
    ci.emplace_back(codec(cfg.pad), codec(cfg.pad) + H);
    ci.emplace_back(codec(cfg.bos), codec(cfg.bos) + H);
    const int L = (int)ci.size();
    prompt.clear();
    // 7b. instruct embeddings are prepended (generation.rs:613-622)
    if (!r.instruct_ids.empty()) {
        std::vector<float> ie(r.instruct_ids.size() * (size_t)H);
        embed_and_project_text(r.instruct_ids.data(), (int)r.instruct_ids.size(), ie.data());
        prompt.insert(prompt.end(), ie.begin(), ie.end());
    }
    // 5. role (first three text tokens)
    if (r.text_ids.size() < 4) throw std::runtime_error("oracle: text_ids too short");
    std::vector<float> role((size_t)3 * H);
    embed_and_project_text(r.text_ids.data(), 3, role.data());
    prompt.insert(prompt.end(), role.begin(), role.end());
    // 6. tts layer (pad x (L-2), bos) + codec rows 0..L-2
    for (int i = 0; i < L - 1; ++i) push(prompt, i < L - 2 ? tts_pad.data() : tts_bos, ci[(size_t)i].data());
    const int n_text = (int)r.text_ids.size();

    if (r.ref_T > 0 && !r.ref_text_ids.empty()) {  // 8. ICL (generation.rs:629-696, 292-416)
        int tce = n_text >= 5 ? n_text - 5 : 0;
        std::vector<uint32_t> tc = tce > 3 ? std::vector<uint32_t>(r.text_ids.begin() + 3, r.text_ids.begin() + tce)
                                           : std::vector<uint32_t>(r.text_ids.begin() + 3, r.text_ids.begin() + 4);
        int n_ref = (int)r.ref_text_ids.size();
        int rte = n_ref >= 2 ? n_ref - 2 : 0;
        std::vector<uint32_t> rc = rte > 3 ? std::vector<uint32_t>(r.ref_text_ids.begin() + 3, r.ref_text_ids.begin() + rte)
                                           : std::vector<uint32_t>(r.ref_text_ids.begin() + 3, r.ref_text_ids.begin() + 4);
        std::vector<uint32_t> comb(rc);
        comb.insert(comb.end(), tc.begin(), tc.end());
        std::vector<float> te(comb.size() * (size_t)H);
        embed_and_project_text(comb.data(), (int)comb.size(), te.data());
        te.insert(te.end(), tts_eos, tts_eos + H);
        const int text_lens = (int)comb.size() + 1;
        // codec embed: bos, then per-frame sum over codebooks (generation.rs:242-273)
        std::vector<float> ce(codec(cfg.bos), codec(cfg.bos) + H);
        for (int t = 0; t < r.ref_T; ++t) {
            std::vector<float> s((size_t)H);
            sum_code_embeddings(&r.ref_codes[(size_t)t * cfg.groups], s.data());
            ce.insert(ce.end(), s.begin(), s.end());
        }
        const int codec_lens = r.ref_T + 1;
        if (r.non_streaming) {
            for (int i = 0; i < text_lens; ++i) push(prompt, &te[(size_t)i * H], codec(cfg.pad));
            for (int i = 0; i < codec_lens; ++i) push(prompt, &ce[(size_t)i * H], tts_pad.data());
            trailing = tts_pad;
            Tt = 1;
        } else if (text_lens > codec_lens) {
            for (int i = 0; i < codec_lens; ++i) push(prompt, &te[(size_t)i * H], &ce[(size_t)i * H]);
            trailing.assign(te.begin() + (size_t)codec_lens * H, te.end());
            Tt = text_lens - codec_lens;
        } else {
            for (int i = 0; i < codec_lens; ++i)
                push(prompt, i < text_lens ? &te[(size_t)i * H] : tts_pad.data(), &ce[(size_t)i * H]);
            trailing = tts_pad;
            Tt = 1;
        }
        P = (int)(prompt.size() / (size_t)H);
        return;
    }

    if (r.non_streaming && n_text > 8) {  // 9. non-streaming tail (generation.rs:699-747)
        const int end = n_text - 5;
        const int n = end - 3;
        std::vector<float> te((size_t)n * H);
        embed_and_project_text(r.text_ids.data() + 3, n, te.data());
        for (int i = 0; i < n; ++i) push(prompt, &te[(size_t)i * H], codec(cfg.pad));
        push(prompt, tts_eos, codec(cfg.pad));
        push(prompt, tts_pad.data(), codec(cfg.bos));
        trailing = tts_pad;
        Tt = 1;
    } else {  // streaming (generation.rs:749-789)
        std::vector<float> fe((size_t)H);
        embed_and_project_text(r.text_ids.data() + 3, 1, fe.data());
        push(prompt, fe.data(), ci[(size_t)L - 1].data());
        if (n_text > 9) {
            const int n = n_text - 5 - 4;
            trailing.assign((size_t)n * H, 0.f);
            embed_and_project_text(r.text_ids.data() + 4, n, trailing.data());
            trailing.insert(trailing.end(), tts_eos, tts_eos + H);
            Tt = n + 1;
        } else {
            trailing = tts_pad;
            Tt = 1;
        }
    }
    P = (int)(prompt.size() / (size_t)H);
}

void Model::talker_forward(float* x, int T, int pos0, std::vector<KVCacheLayer>& cache, std::vector<float>& hidden) const {
    for (int l = 0; l < cfg.talker.layers; ++l)
        decoder_layer(w, "talker.model.layers." + std::to_string(l) + ".", cfg.talker, rope_t, x, T, pos0, MASK_CAUSAL,
                      cache[(size_t)l]);
    hidden.resize((size_t)T * cfg.talker.hidden);
    rmsnorm(x, W(w, "talker.model.norm.weight").p(), hidden.data(), T, cfg.talker.hidden, cfg.talker.eps);
}

void Model::talker_logits(const float* h, float* logits) const {
    linear(h, W(w, "talker.codec_head.weight").p(), nullptr, logits, 1, cfg.vocab, cfg.talker.hidden);
}

static float top2_gap(const float* l, int64_t n) {
    float a = -INFINITY, b = -INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        if (l[i] > a) { b = a; a = l[i]; }
        else if (l[i] > b) b = l[i];
    }
    return a - b;
}

void Model::code_predictor(const float* talker_hidden, int64_t cb0, const SamplingConfig& sub, MT19937* rng,
                           uint32_t* out_codes, float* min_margin, std::vector<float>* logits_dbg) const {
    const int H = cfg.talker.hidden, Hc = cfg.cp.hidden, G = cfg.groups, Vc = cfg.cp_vocab;
    const bool proj = has(w, "talker.code_predictor.small_to_mtp_projection.weight");
    auto project = [&](const float* in, int T, float* out) {
        if (proj)
            linear(in, W(w, "talker.code_predictor.small_to_mtp_projection.weight").p(),
                   W(w, "talker.code_predictor.small_to_mtp_projection.bias").p(), out, T, Hc, H);
        else std::memcpy(out, in, sizeof(float) * (size_t)T * H);
    };
    std::vector<KVCacheLayer> cache((size_t)cfg.cp.layers);
    for (auto& c : cache) {
        c.cap = G + 1;
        c.k.assign((size_t)cfg.cp.nkv * c.cap * cfg.cp.d, 0.f);
        c.v.assign((size_t)cfg.cp.nkv * c.cap * cfg.cp.d, 0.f);
    }
    // pass 0: two tokens, NO mask (code_predictor.rs:455-460 + attention/mod.rs:413-417)
    std::vector<float> in((size_t)2 * H), x((size_t)2 * Hc), hn((size_t)2 * Hc), logits((size_t)Vc);
    std::memcpy(&in[0], talker_hidden, sizeof(float) * H);
    std::memcpy(&in[(size_t)H], W(w, "talker.model.codec_embedding.weight").p() + (size_t)cb0 * H, sizeof(float) * H);
    project(in.data(), 2, x.data());
    for (int l = 0; l < cfg.cp.layers; ++l)
        decoder_layer(w, "talker.code_predictor.model.layers." + std::to_string(l) + ".", cfg.cp, rope_c, x.data(), 2, 0,
                      MASK_NONE, cache[(size_t)l]);
    rmsnorm(x.data(), W(w, "talker.code_predictor.model.norm.weight").p(), hn.data(), 2, Hc, cfg.cp.eps);
    const std::vector<int64_t> nohist;
    auto head_sample = [&](int j, const float* hrow) -> uint32_t {
        linear(hrow, W(w, "talker.code_predictor.lm_head." + std::to_string(j) + ".weight").p(), nullptr, logits.data(), 1, Vc, Hc);
        if (logits_dbg) logits_dbg->insert(logits_dbg->end(), logits.begin(), logits.end());
        if (min_margin && !sub.do_sample) *min_margin = std::min(*min_margin, top2_gap(logits.data(), Vc));
        return (uint32_t)sample_token(logits.data(), Vc, sub, nohist, rng);
    };
    out_codes[0] = head_sample(0, &hn[(size_t)Hc]);
    int pos = 2;
    for (int idx = 2; idx < G; ++idx) {  // code_predictor.rs:480-527
        const float* e = W(w, "talker.code_predictor.model.codec_embedding." + std::to_string(idx - 2) + ".weight").p() +
                         (size_t)out_codes[idx - 2] * H;
        project(e, 1, x.data());
        for (int l = 0; l < cfg.cp.layers; ++l)
            decoder_layer(w, "talker.code_predictor.model.layers." + std::to_string(l) + ".", cfg.cp, rope_c, x.data(), 1, pos,
                          MASK_CAUSAL, cache[(size_t)l]);
        rmsnorm(x.data(), W(w, "talker.code_predictor.model.norm.weight").p(), hn.data(), 1, Hc, cfg.cp.eps);
        out_codes[idx - 1] = head_sample(idx - 1, hn.data());
        ++pos;
    }
}

void Model::generate(const GenRequest& r, const SamplingConfig& s_in, int max_new_tokens, MT19937* rng, bool trace,
                     GenOutput& out) const {
    const int H = cfg.talker.hidden, G = cfg.groups, V = cfg.vocab;
    SamplingConfig s = s_in;
    s.eos_token_id = cfg.eos;  // generation.rs:841-853
    s.suppress_lo = V >= 1024 ? V - 1024 : 0;
    s.suppress_hi = V;
    const SamplingConfig sub = s.for_subtalker();
    std::vector<float> prompt, trailing, tts_pad;
    int P = 0, Tt = 0;
    build_prompt(r, prompt, P, trailing, Tt, tts_pad);
    std::vector<KVCacheLayer> cache((size_t)cfg.talker.layers);
    for (auto& c : cache) {
        c.cap = P + max_new_tokens + 1;
        c.k.assign((size_t)cfg.talker.nkv * c.cap * cfg.talker.d, 0.f);
        c.v.assign((size_t)cfg.talker.nkv * c.cap * cfg.talker.d, 0.f);
    }
    std::vector<float> hidden, logits((size_t)V), next((size_t)H);
    talker_forward(prompt.data(), P, 0, cache, hidden);
    const float* hlast = &hidden[(size_t)(P - 1) * H];
    std::vector<int64_t> hist;
    out = GenOutput();
    int step = 0;
    int eos_step = -1;
    std::vector<float> hcur(hlast, hlast + H);
    auto frame = [&]() -> bool {  // returns false when the very first token is EOS (generation.rs:939)
        talker_logits(hcur.data(), logits.data());
        if (trace) {
            out.logits_trace.insert(out.logits_trace.end(), logits.begin(), logits.end());
            out.hidden_trace.insert(out.hidden_trace.end(), hcur.begin(), hcur.end());
        }
        if (!s.do_sample) {
            std::vector<float> l(logits);
            apply_repetition_penalty(l, hist, s.repetition_penalty);
            for (int i = s.suppress_lo; i < s.suppress_hi; ++i)
                if (i != s.eos_token_id) l[(size_t)i] = -INFINITY;
            if ((int)hist.size() < s.min_new_tokens) l[(size_t)s.eos_token_id] = -INFINITY;
            out.min_margin = std::min(out.min_margin, top2_gap(l.data(), V));
        }
        int64_t tok = sample_token(logits.data(), V, s, hist, rng);
        hist.push_back(tok);
        if (eos_step < 0 && tok == cfg.eos) eos_step = step;
        if (step == 0 && tok == cfg.eos) return false;
        std::vector<uint32_t> codes((size_t)G);
        codes[0] = (uint32_t)tok;
        code_predictor(hcur.data(), tok, sub, rng, &codes[1], &out.min_margin, nullptr);
        out.codes.insert(out.codes.end(), codes.begin(), codes.end());
        sum_code_embeddings(codes.data(), next.data());
        const float* add = step < Tt ? &trailing[(size_t)step * H] : tts_pad.data();  // generation.rs:971-978
        for (int i = 0; i < H; ++i) next[(size_t)i] = next[(size_t)i] + add[i];
        return true;
    };
    if (max_new_tokens > 0 && frame()) {
        step = 1;
        while (step < max_new_tokens) {
            if (eos_step >= 0) break;  // batch 1: all samples hit EOS
            std::vector<float> x(next);
            talker_forward(x.data(), 1, cache[0].len, cache, hidden);
            hcur.assign(hidden.begin(), hidden.begin() + H);
            frame();
            ++step;
        }
    }
    out.num_steps = step;
    out.effective_len = eos_step >= 0 ? eos_step : step;
}

}  // namespace orc
