// Shared layer runner: one pre-norm transformer layer / stack over M tokens in flight, used by the talker,
// the code predictor and the codec's sliding-window transformer.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <vector>

#include "model.hpp"

namespace qtts {

struct Workspace {
    DevBuf x, xn, part, q, attn, act;  // residual f32 | gemm input T | GEMM output f32 | q f32 | attn T | mlp act T
    // fused decode path: reduced GEMM output f32 | x * norm_w (T) | per-row sum-of-squares partials (two generations)
    DevBuf out, xb, ssA, ssB;
    long ss_ld = 0;  // row stride of ssA / ssB
};

struct GemmTimer {
    bool enabled = false;
    std::vector<hipEvent_t> ev;
    std::vector<int> shape;  // (M, N, K) per timed launch
    size_t used = 0;
    double bytes = 0;
    ~GemmTimer() {
        for (auto e : ev) (void)hipEventDestroy(e);
    }
    // per-shape totals: rows of (M, N, K, launches, total microseconds)
    std::vector<double> by_shape() const {
        std::vector<double> rows;
        for (size_t i = 0; i + 1 < used; i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) != hipSuccess) continue;
            const int* sh = &shape[(i / 2) * 3];
            size_t r = 0;
            for (; r < rows.size(); r += 5)
                if (rows[r] == sh[0] && rows[r + 1] == sh[1] && rows[r + 2] == sh[2]) break;
            if (r == rows.size()) rows.insert(rows.end(), {(double)sh[0], (double)sh[1], (double)sh[2], 0.0, 0.0});
            rows[r + 3] += 1.0;
            rows[r + 4] += (double)t * 1e3;
        }
        return rows;
    }
    double total_ms() const {
        double ms = 0;
        for (size_t i = 0; i + 1 < used; i += 2) {
            float t = 0;
            if (hipEventElapsedTime(&t, ev[i], ev[i + 1]) == hipSuccess) ms += t;
        }
        return ms;
    }
};

struct Runner {
    Model& m;
    Workspace& g;
    GemmTimer* timer;
    hipStream_t st;
    int dt;
    size_t es;
    // from this many tokens on the large-M kernel (no split-K slabs) is used; the slab workspaces are sized for at most 512 rows
    static int rows_kernel_min_m() {
        static const int v = getenv("QTTS_ROWS_MIN_M") ? std::min(513, std::max(17, atoi(getenv("QTTS_ROWS_MIN_M")))) : 513;
        return v;
    }

    Runner(Model& model, Workspace& ws, GemmTimer* t) : m(model), g(ws), timer(t), st(model.ctx->stream), dt(model.dtype), es(dtype_size(model.dtype)) {}

    // split-K factor the consumers of g.part must assume for a GEMM of this shape
    int ks_for(int M, int N, int K) const {
        if (M >= rows_kernel_min_m() && N % 64 == 0) return 1;
        return gemm_pick_ks(M, N, K, dt);
    }
    long kstride(int M, int N) const { return (long)gemm_pad_rows(M) * N; }

    void gemm(const void* W, const void* X, int M, int N, int K, int KS, int ldx) {
        const bool rows = M >= rows_kernel_min_m() && N % 64 == 0;
        const bool time = timer && timer->enabled && timer->used + 2 <= 400000;
        if (time) {
            while (timer->ev.size() < timer->used + 2) {
                hipEvent_t e;
                QTTS_HIP(hipEventCreate(&e));
                timer->ev.push_back(e);
            }
            QTTS_HIP(hipEventRecord(timer->ev[timer->used], st));
        }
        if (rows) launch_gemm_rows(st, dt, W, X, g.part.as<float>(), M, N, K, ldx, 1, 0);
        else launch_gemm(st, dt, W, X, g.part.as<float>(), M, N, K, KS, ldx);
        if (time) {
            QTTS_HIP(hipEventRecord(timer->ev[timer->used + 1], st));
            timer->used += 2;
            timer->bytes += (double)N * K * es;
            timer->shape.push_back(M);
            timer->shape.push_back(N);
            timer->shape.push_back(K);
        }
    }

    // One decoder layer over M tokens (nn/decoder_layer.rs:63-86; codec: audio/tokenizer/v2/transformer.rs:279-300).
    // On entry x holds the residual stream and xn = RMSNorm_ln1(x); on exit xn = RMSNorm_next(x) with `next_norm`
    // the following layer's input norm (or the stack's final norm).
    // `decode`: every token's key range ends at its own position and tokens of the launch do not attend to each other
    void layer(const StackW& s, int l, int M, const Tok* toks, const KVView& kv, int kv_layer, const float* next_norm, void* xn_out,
               float* xn_f32_out, bool decode = false) {
        const StackCfg& c = s.cfg;
        const LayerW& w = s.layers[(size_t)l];
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        int ks = ks_for(M, nqkv, c.hidden);
        gemm(w.qkv, g.xn.p, M, nqkv, c.hidden, ks, c.hidden);
        QkvPostArgs qa{};
        qa.part = g.part.as<float>(); qa.KS = ks; qa.ks_stride = kstride(M, nqkv); qa.bias = w.qkv_bias;
        qa.q_norm = w.q_norm; qa.k_norm = w.k_norm; qa.eps = c.eps; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
        qa.rope_mode = 0; qa.toks = toks; qa.M = M; qa.nh = c.nh; qa.nkv = c.nkv; qa.d = c.d; qa.layer = kv_layer;
        qa.q_out = g.q.as<float>();
        if (decode) {
            launch_attention_decode(st, dt, qa, kv, (float)std::pow((double)c.d, -0.5), g.attn.p);
        } else {
            launch_qkv_post(st, dt, qa, kv);
            launch_attention(st, dt, g.q.as<float>(), toks, kv, kv_layer, M, c.nh, (float)std::pow((double)c.d, -0.5), g.attn.p);
        }
        ks = ks_for(M, c.hidden, nq);
        gemm(w.o, g.attn.p, M, c.hidden, nq, ks, nq);
        launch_resid_norm(st, dt, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), w.o_bias, w.ls_attn, w.ln2, c.eps,
                          g.x.as<float>(), g.xn.p, nullptr, M, c.hidden);
        ks = ks_for(M, 2 * c.inter, c.hidden);
        gemm(w.gate_up, g.xn.p, M, 2 * c.inter, c.hidden, ks, c.hidden);
        launch_swiglu(st, dt, g.part.as<float>(), ks, kstride(M, 2 * c.inter), g.act.p, M, c.inter);
        ks = ks_for(M, c.hidden, c.inter);
        gemm(w.down, g.act.p, M, c.hidden, c.inter, ks, c.inter);
        launch_resid_norm(st, dt, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), nullptr, w.ls_mlp, next_norm, c.eps,
                          g.x.as<float>(), xn_out, xn_f32_out, M, c.hidden);
    }

    // ---- fused decode path (k_gemm_fused.hip): the residual stream is (x f32, xb = T(x * next norm weight), ssA partials)
    void gemm_fused(const void* W, const void* X, int ldx, int M, int N, int K, GemmEpi e) {
        e.ss_ld = g.ss_ld;
        const bool time = timer && timer->enabled && timer->used + 2 <= 400000;
        if (time) {
            while (timer->ev.size() < timer->used + 2) {
                hipEvent_t ev;
                QTTS_HIP(hipEventCreate(&ev));
                timer->ev.push_back(ev);
            }
            QTTS_HIP(hipEventRecord(timer->ev[timer->used], st));
        }
        launch_gemm_fused(st, dt, W, X, ldx, M, N, K, e);
        if (time) {
            QTTS_HIP(hipEventRecord(timer->ev[timer->used + 1], st));
            timer->used += 2;
            timer->bytes += (double)N * K * es;
            timer->shape.push_back(M);
            timer->shape.push_back(N);
            timer->shape.push_back(K);
        }
    }

    // One decoder layer on the fused path.  In: x, xb = T(x * ln1), ssA (nt_in tiles).  Out: x, xb = T(x * next_norm), ssA;
    // returns the tile count of the new ssA.
    int layer_fused(const StackW& s, int l, int M, const Tok* toks, const KVView& kv, int kv_layer, const float* next_norm, int nt_in,
                    bool decode) {
        const StackCfg& c = s.cfg;
        const LayerW& w = s.layers[(size_t)l];
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        GemmEpi e{};
        e.mode = GEMM_EPI_PLAIN; e.ss_in = g.ssA.as<float>(); e.ntiles_in = nt_in; e.H_in = c.hidden; e.eps = c.eps; e.bias = w.qkv_bias;
        e.out_f32 = g.out.as<float>(); e.ld_out = nqkv;
        gemm_fused(w.qkv, g.xb.p, c.hidden, M, nqkv, c.hidden, e);
        QkvPostArgs qa{};
        qa.part = g.out.as<float>(); qa.KS = 1; qa.ks_stride = 0; qa.bias = nullptr;
        qa.q_norm = w.q_norm; qa.k_norm = w.k_norm; qa.eps = c.eps; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
        qa.rope_mode = 0; qa.toks = toks; qa.M = M; qa.nh = c.nh; qa.nkv = c.nkv; qa.d = c.d; qa.layer = kv_layer;
        qa.q_out = g.q.as<float>();
        const float scale = (float)std::pow((double)c.d, -0.5);
        if (decode) {
            launch_attention_decode(st, dt, qa, kv, scale, g.attn.p);
        } else {
            launch_qkv_post(st, dt, qa, kv);
            launch_attention(st, dt, g.q.as<float>(), toks, kv, kv_layer, M, c.nh, scale, g.attn.p);
        }
        GemmEpi eo{};
        eo.mode = GEMM_EPI_RESID; eo.bias = w.o_bias; eo.resid = g.x.as<float>(); eo.x_out = g.x.as<float>(); eo.xb_out = g.xb.p;
        eo.nw = w.ln2; eo.ss_out = g.ssB.as<float>();
        gemm_fused(w.o, g.attn.p, nq, M, c.hidden, nq, eo);
        const int nt_b = gemm_fused_ntiles(M, c.hidden, GEMM_EPI_RESID);
        GemmEpi eg{};
        eg.mode = GEMM_EPI_SWIGLU; eg.ss_in = g.ssB.as<float>(); eg.ntiles_in = nt_b; eg.H_in = c.hidden; eg.eps = c.eps; eg.act = g.act.p;
        gemm_fused(w.gate_up_il, g.xb.p, c.hidden, M, 2 * c.inter, c.hidden, eg);
        GemmEpi ed{};
        ed.mode = GEMM_EPI_RESID; ed.resid = g.x.as<float>(); ed.x_out = g.x.as<float>(); ed.xb_out = g.xb.p; ed.nw = next_norm;
        ed.ss_out = g.ssA.as<float>();
        gemm_fused(w.down, g.act.p, c.inter, M, c.hidden, c.inter, ed);
        return gemm_fused_ntiles(M, c.hidden, GEMM_EPI_RESID);
    }

    // whole stack on the fused path; afterwards xb = T(x * final_norm) and ssA holds the returned number of tiles
    int stack_fused(const StackW& s, int M, const Tok* toks, const KVView& kv, int nt_in, bool kv_per_layer, bool decode) {
        int nt = nt_in;
        for (int l = 0; l < s.cfg.layers; ++l) {
            const bool last = l + 1 == s.cfg.layers;
            nt = layer_fused(s, l, M, toks, kv, kv_per_layer ? l : 0, last ? s.final_norm : s.layers[(size_t)l + 1].ln1, nt, decode);
        }
        return nt;
    }

    // kv_per_layer: the KV view holds every layer (decode caches); otherwise one layer's worth is reused
    void stack(const StackW& s, int M, const Tok* toks, const KVView& kv, void* final_xn, float* final_xn_f32, bool kv_per_layer = true,
               bool decode = false) {
        for (int l = 0; l < s.cfg.layers; ++l) {
            const bool last = l + 1 == s.cfg.layers;
            layer(s, l, M, toks, kv, kv_per_layer ? l : 0, last ? s.final_norm : s.layers[(size_t)l + 1].ln1, last ? final_xn : g.xn.p,
                  last ? final_xn_f32 : nullptr, decode);
        }
    }
};

}  // namespace qtts
