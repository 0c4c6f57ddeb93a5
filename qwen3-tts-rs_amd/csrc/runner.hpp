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
    // bf16 runs: xn, attn and act carry a remainder plane this many elements behind the first (0 = single plane),
    // so that the GEMMs see their activations to about 16 significant bits; the buffers are then allocated twice as large
    long lo_xn = 0, lo_attn = 0, lo_act = 0;
    // fused decode stacks: per-row partial sums of squares of the residual stream, two generations, [tiles][ss_ld]
    DevBuf ssA, ssB;
    long ss_ld = 0;
    // 64-row decode passes through the fragment-order GEMM (k_gemm_frag.hip): its inputs in MFMA fragment order, 64 rows each
    DevBuf xnp, attnp, actp;
    // allocates the three GEMM-input buffers for `rows` rows of `wx`, `wq`, `wa` elements (split when `split`)
    void alloc_inputs(size_t rows, size_t wx, size_t wq, size_t wa, size_t es, bool split) {
        const size_t f = split ? 2 : 1;
        xn.ensure(rows * wx * es * f);
        attn.ensure(rows * wq * es * f);
        act.ensure(rows * wa * es * f);
        lo_xn = split ? (long)(rows * wx) : 0;
        lo_attn = split ? (long)(rows * wq) : 0;
        lo_act = split ? (long)(rows * wa) : 0;
    }
};

// Activations of bf16 runs are split into a bf16 value and a bf16 remainder.  QTTS_SPLIT_ACT: 0 = never (one MFMA pass
// fewer per GEMM; logits then carry bf16 rounding noise of about 2e-2 max-abs at the 28-layer shapes), 1 (default) = the
// talker and the codec's low-rate stages, 2 = the five-layer code predictor too.
inline int split_level(int dtype) {
    static const int level = getenv("QTTS_SPLIT_ACT") ? atoi(getenv("QTTS_SPLIT_ACT")) : 1;
    return dtype == BF16 ? level : 0;
}
inline bool split_activations(int dtype) { return split_level(dtype) >= 1; }

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
    // decode launches whose tokens all sit at one position (row m = token m, keys 0 .. uni_len - 1) say so here: the attention
    // kernel then takes the description from its arguments instead of loading it (0 = read the token table)
    int uni_len = 0;
    // g.xn holds T(x * norm weight) whose 1/rms is still to be applied: this many slices of partial sums of squares are in ssA
    // (0: xn is normalised).  Set and consumed by consecutive layer() calls of one stack.
    int xn_tiles = 0;
    // prefill-style launches (decode == false) whose tokens come in runs of one cache row: the launch's tokens cut into tiles
    // (device array, build_attn_tiles) send their attention through the matrix-core tile kernel; null = one block per token
    const AttnTile* tiles = nullptr;
    int n_tiles = 0;
    Model& m;
    Workspace& g;
    GemmTimer* timer;
    hipStream_t st;
    int dt;
    size_t es;
    // remainder-plane offsets the layers use for xn / attn / act: the workspace's, or 0 after split(false) (a stack whose
    // depth keeps bf16 rounding within the tolerance runs single-plane: one MFMA pass and one X tile fewer per GEMM)
    long lo_xn, lo_attn, lo_act;
    void split(bool on) {
        lo_xn = on ? g.lo_xn : 0;
        lo_attn = on ? g.lo_attn : 0;
        lo_act = on ? g.lo_act : 0;
    }
    // from this many tokens on the large-M kernel (no split-K slabs) is used; the slab workspaces are sized for at most 512 rows
    static int rows_kernel_min_m() {
        static const int v = getenv("QTTS_ROWS_MIN_M") ? std::min(513, std::max(17, atoi(getenv("QTTS_ROWS_MIN_M")))) : 513;
        return v;
    }

    // Admission work (text projection, prompt prefill) sends every GEMM through the large-M kernel, whatever the token count:
    // that kernel walks K in one fixed order per output element, so an utterance's prompt state does not depend on how many
    // other prompts were admitted with it (the split-K plans of the small-M kernels change the summation order with M).
    bool canonical_rows = false;
    bool rows_kernel(int M, int N) const { return (M >= rows_kernel_min_m() || canonical_rows) && N % 64 == 0; }

    bool tune_shared;  // decode kernels picked for several streams in flight (the weight set's choice, made once: kernels.hpp)
    Runner(Model& model, Workspace& ws, GemmTimer* t) : m(model), g(ws), timer(t), st(model.ctx->stream), dt(model.dtype), es(dtype_size(model.dtype)) {
        split(true);
        tune_shared = model.tuning->peek() == TUNE_SHARED;
    }

    // split-K factor the consumers of g.part must assume for a GEMM of this shape
    int ks_for(int M, int N, int K) const {
        if (rows_kernel(M, N)) return 1;
        return gemm_pick_ks(M, N, K, dt);
    }
    long kstride(int M, int N) const { return (long)gemm_pad_rows(M) * N; }

    void gemm(WMat W, const void* X, int M, int N, int K, int KS, int ldx, long x_lo = 0) {
        const bool rows = rows_kernel(M, N);
        const bool time = timer && timer->enabled && timer->used + 2 <= 400000;
        if (time) {
            while (timer->ev.size() < timer->used + 2) {
                hipEvent_t e;
                QTTS_HIP(hipEventCreate(&e));
                timer->ev.push_back(e);
            }
            QTTS_HIP(hipEventRecord(timer->ev[timer->used], st));
        }
        if (rows) launch_gemm_rows(st, dt, W, X, g.part.as<float>(), M, N, K, ldx, 1, 0, x_lo);
        else launch_gemm(st, dt, W, X, g.part.as<float>(), M, N, K, KS, ldx, x_lo);
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
    // xn_out_lo: remainder-plane offset of xn_out (lo_xn when xn_out is g.xn)
    void layer(const StackW& s, int l, int M, const Tok* toks, const KVView& kv, int kv_layer, const float* next_norm, void* xn_out,
               float* xn_f32_out, bool decode, long xn_out_lo) {
        const StackCfg& c = s.cfg;
        const LayerW& w = s.layers[(size_t)l];
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        // decode steps of bf16 stacks: q|k|v and gate|up own finished outputs (no split-K slabs; SwiGLU inside the GEMM)
        const bool fk = decode && full_k_usable(s, M);
        // decode steps of a bf16 stack: the two residual / norm passes leave xn = T(x * w) and per-slice sums of squares, and the
        // GEMM that reads xn applies 1/rms (one wave per 256 columns instead of a block-wide reduction per row)
        const bool pn = fk && partial_norm_enabled() && !w.ls_attn && !w.ls_mlp && c.hidden % 256 == 0 && g.ssA.p && g.ssB.p &&
                        g.ss_ld >= fused_rows(M);
        QTTS_CHECK(xn_tiles == 0 || fk, ERR_STATE, "layer: the input's norm scale is pending but this path cannot apply it");
        int ks = fk ? 1 : ks_for(M, nqkv, c.hidden);
        if (fk) {
            FusedEpi e{};
            e.mode = FUSED_PLAIN; e.bias = w.qkv_bias; e.out_f32 = g.part.as<float>(); e.ld_out = nqkv;
            if (xn_tiles) { e.ss_in = g.ssA.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = xn_tiles; e.H_in = c.hidden; e.eps = c.eps; }
            xn_tiles = 0;
            timed(M, nqkv, c.hidden, [&] { wide_gemm(w.qkv, g.xn.p, c.hidden, lo_xn, M, nqkv, c.hidden, e); });
        } else {
            gemm(w.qkv, g.xn.p, M, nqkv, c.hidden, ks, c.hidden, lo_xn);
        }
        QkvPostArgs qa{};
        qa.part = g.part.as<float>(); qa.KS = ks; qa.ks_stride = kstride(M, nqkv); qa.bias = fk ? nullptr : w.qkv_bias;
        qa.q_norm = w.q_norm; qa.k_norm = w.k_norm; qa.eps = c.eps; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
        qa.rope_mode = 0; qa.toks = toks; qa.M = M; qa.nh = c.nh; qa.nkv = c.nkv; qa.d = c.d; qa.layer = kv_layer;
        qa.q_out = g.q.as<float>(); qa.uni_len = decode ? uni_len : 0; qa.small = tune_shared ? 1 : 0;
        if (decode) {
            launch_attention_decode(st, dt, qa, kv, (float)std::pow((double)c.d, -0.5), g.attn.p, lo_attn);
        } else {
            launch_qkv_post(st, dt, qa, kv);
            attention(c, toks, kv, kv_layer, M, (float)std::pow((double)c.d, -0.5));
        }
        // o projection: in decode steps of a bf16 stack one launch adds the residual and leaves xn = T(x * ln2) with the per-tile
        // sums of squares that gate|up turns into the row scale (instead of split-K slabs + a residual / norm pass)
        const bool o_fused = fk && !w.ls_attn && o_fused_enabled() && g.ssB.p && g.ss_ld >= fused_rows(M) && c.hidden % 16 == 0 && nq % 64 == 0;
        int nt_b = 0;
        if (o_fused) {
            nt_b = k_split_resid(w.o, g.attn.p, lo_attn, M, c.hidden, nq, w.o_bias, true, w.ln2, g.ssB.as<float>());
        } else {
            ks = ks_for(M, c.hidden, nq);
            gemm(w.o, g.attn.p, M, c.hidden, nq, ks, nq, lo_attn);
            if (pn)
                nt_b = launch_resid_partial(st, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), w.o_bias, w.ln2, g.x.as<float>(),
                                            g.xn.p, lo_xn, g.ssB.as<float>(), g.ss_ld, M, c.hidden);
            else
                launch_resid_norm(st, dt, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), w.o_bias, w.ls_attn, w.ln2, c.eps,
                                  g.x.as<float>(), g.xn.p, nullptr, M, c.hidden, lo_xn);
        }
        if (fk) {
            FusedEpi e{};
            e.mode = FUSED_SWIGLU; e.act = g.act.p; e.act_lo = lo_act;
            if (nt_b) { e.ss_in = g.ssB.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = nt_b; e.H_in = c.hidden; e.eps = c.eps; }
            timed(M, 2 * c.inter, c.hidden, [&] { wide_gemm(w.gate_up_il, g.xn.p, c.hidden, lo_xn, M, 2 * c.inter, c.hidden, e); });
        } else {
            ks = ks_for(M, 2 * c.inter, c.hidden);
            gemm(w.gate_up, g.xn.p, M, 2 * c.inter, c.hidden, ks, c.hidden, lo_xn);
            launch_swiglu(st, dt, g.part.as<float>(), ks, kstride(M, 2 * c.inter), g.act.p, M, c.inter, lo_act);
        }
        ks = ks_for(M, c.hidden, c.inter);
        gemm(w.down, g.act.p, M, c.hidden, c.inter, ks, c.inter, lo_act);
        if (pn && xn_out == g.xn.p && !xn_f32_out && xn_out_lo == lo_xn) {  // the next layer of the stack reads xn: it applies the scale
            xn_tiles = launch_resid_partial(st, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), nullptr, next_norm, g.x.as<float>(),
                                            g.xn.p, lo_xn, g.ssA.as<float>(), g.ss_ld, M, c.hidden);
        } else {
            launch_resid_norm(st, dt, g.x.as<float>(), g.part.as<float>(), ks, kstride(M, c.hidden), nullptr, w.ls_mlp, next_norm, c.eps,
                              g.x.as<float>(), xn_out, xn_f32_out, M, c.hidden, xn_out_lo);
        }
    }
    // attention of a launch whose keys and values are all in the cache: tiles on the matrix cores when the caller described
    // the launch's runs (QTTS_ATTN_TILES=0: always one block per token and kv head)
    void attention(const StackCfg& c, const Tok* toks, const KVView& kv, int kv_layer, int M, float scale) {
        static const bool off = getenv("QTTS_ATTN_TILES") && atoi(getenv("QTTS_ATTN_TILES")) == 0;
        if (tiles && n_tiles > 0 && !off && attention_tiles_supported(c.d, c.nh, c.nkv))
            launch_attention_tiles(st, dt, g.q.as<float>(), toks, tiles, n_tiles, kv, kv_layer, c.nh, scale, g.attn.p, lo_attn);
        else
            launch_attention(st, dt, g.q.as<float>(), toks, kv, kv_layer, M, c.nh, scale, g.attn.p, lo_attn);
    }
    // QTTS_PARTIAL_NORM=0: residual + RMSNorm passes of the decode layers as one block per row again
    static bool partial_norm_enabled() {
        static const bool off = getenv("QTTS_PARTIAL_NORM") && atoi(getenv("QTTS_PARTIAL_NORM")) == 0;
        return !off;
    }

    // QTTS_O_FUSED=1 switches the one-launch o projection on.  Measured on the 1.7B talker at 64 rows: 28 launches fewer per
    // frame and the same single-stream frame time (4.99 ms either way: gate|up then sums 128 tiles of partials per row), but
    // three sessions in flight lose 4 % (1698 against 1764 audio-s/s) to its 512 blocks of eight waves - so it stays off.
    static bool o_fused_enabled() {
        static const bool on = getenv("QTTS_O_FUSED") && atoi(getenv("QTTS_O_FUSED")) == 1;
        return on;
    }
    // q|k|v and gate|up of this stack can go through the full-K kernel for M tokens (QTTS_FULL_K=0: never)
    bool full_k_usable(const StackW& s, int M) const {
        static const bool off = getenv("QTTS_FULL_K") && atoi(getenv("QTTS_FULL_K")) == 0;
        const StackCfg& c = s.cfg;
        if (off || dt != BF16 || M > 512 || s.layers.empty() || !s.layers[0].gate_up_il) return false;
        return full_k_fits(c.hidden, lo_xn != 0) && ((c.nh + 2 * c.nkv) * c.d) % 64 == 0 && (2 * c.inter) % 64 == 0;
    }

    // finished-output GEMM with epilogue: K walked in phases inside the block at 64-row multiples, whole-K tile otherwise
    void wide_gemm(WMat W, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e) {
        if (gemm_stream_preferred(M, N, K, x_lo != 0, tune_shared)) launch_gemm_stream(st, W, X, ldx, x_lo, M, N, K, e);
        else launch_gemm_full_k(st, W, X, ldx, x_lo, M, N, K, e);
    }

    // ---- fused decode layers (k_gemm_fused.hip): the residual stream is (x f32, xn = T(x * next norm weight), ssA partials)
    template <typename F>
    void timed(int M, int N, int K, F&& launch) {
        const bool time = timer && timer->enabled && timer->used + 2 <= 400000;
        if (time) {
            while (timer->ev.size() < timer->used + 2) {
                hipEvent_t ev;
                QTTS_HIP(hipEventCreate(&ev));
                timer->ev.push_back(ev);
            }
            QTTS_HIP(hipEventRecord(timer->ev[timer->used], st));
        }
        launch();
        if (time) {
            QTTS_HIP(hipEventRecord(timer->ev[timer->used + 1], st));
            timer->used += 2;
            timer->bytes += (double)N * K * es;
            timer->shape.push_back(M);
            timer->shape.push_back(N);
            timer->shape.push_back(K);
        }
    }
    // x = [resid +] X W^T + bias ; xn = T(x * nw) ; partials -> ss_out.  Returns the number of partial tiles.
    int k_split_resid(WMat W, const void* X, long x_lo, int M, int N, int K, const float* bias, bool resid, const float* nw, float* ss_out) {
        FusedEpi e{};
        e.mode = FUSED_RESID; e.ss_ld = g.ss_ld; e.bias = bias; e.resid = resid ? g.x.as<float>() : nullptr; e.x_out = g.x.as<float>();
        e.xb_out = g.xn.p; e.xb_lo = lo_xn; e.nw = nw; e.ss_out = ss_out;
        timed(M, N, K, [&] { launch_gemm_k_split(st, W, X, K, x_lo, M, N, K, e, tune_shared); });
        return fused_resid_tiles(N);
    }
    static bool stack_fusable(const Model& m, const StackW& s) {
        static const bool off = getenv("QTTS_FUSED_STACK") && atoi(getenv("QTTS_FUSED_STACK")) == 0;
        if (off || !fused_gemm_supported(m.dtype, s.cfg.hidden) || s.layers.empty()) return false;
        const StackCfg& c = s.cfg;
        const int nqkv = (c.nh + 2 * c.nkv) * c.d;
        for (const LayerW& w : s.layers)
            if (!w.gate_up_il || w.ls_attn || w.ls_mlp) return false;
        return nqkv % 64 == 0 && (2 * c.inter) % 64 == 0 && c.hidden % 16 == 0 && (c.nh * c.d) % 64 == 0 && c.inter % 64 == 0;
    }
    // One decoder layer, five launches.  In: x, xn = T(x * ln1), ssA (nt_in tiles).  Out: x, xn = T(x * next_norm), ssA.
    // qkv_rows: the layer's q|k|v rows [M][nqkv] f32 are already there (tabulated); the q|k|v GEMM is skipped and xn / ssA are not read
    int layer_fused(const StackW& s, int l, int M, const Tok* toks, const KVView& kv, int kv_layer, const float* next_norm, int nt_in, bool decode,
                    const float* qkv_rows = nullptr) {
        const StackCfg& c = s.cfg;
        const LayerW& w = s.layers[(size_t)l];
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        QTTS_CHECK(lo_xn == 0, ERR_STATE, "fused decode layers take single-plane layer inputs");
        if (!qkv_rows) {
            FusedEpi e{};
            e.mode = FUSED_PLAIN; e.ss_in = g.ssA.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = nt_in; e.H_in = c.hidden; e.eps = c.eps;
            e.bias = w.qkv_bias; e.out_f32 = g.part.as<float>(); e.ld_out = nqkv;
            timed(M, nqkv, c.hidden, [&] { wide_gemm(w.qkv, g.xn.p, c.hidden, 0, M, nqkv, c.hidden, e); });
        }
        QkvPostArgs qa{};
        qa.part = qkv_rows ? qkv_rows : g.part.as<float>(); qa.KS = 1; qa.ks_stride = 0; qa.bias = nullptr;
        qa.q_norm = w.q_norm; qa.k_norm = w.k_norm; qa.eps = c.eps; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
        qa.rope_mode = 0; qa.toks = toks; qa.M = M; qa.nh = c.nh; qa.nkv = c.nkv; qa.d = c.d; qa.layer = kv_layer;
        qa.q_out = g.q.as<float>(); qa.uni_len = decode ? uni_len : 0; qa.small = tune_shared ? 1 : 0;
        const float scale = (float)std::pow((double)c.d, -0.5);
        if (decode) {
            launch_attention_decode(st, dt, qa, kv, scale, g.attn.p, lo_attn);
        } else {
            launch_qkv_post(st, dt, qa, kv);
            attention(c, toks, kv, kv_layer, M, scale);
        }
        const int nt_b = k_split_resid(w.o, g.attn.p, lo_attn, M, c.hidden, nq, w.o_bias, true, w.ln2, g.ssB.as<float>());
        {
            FusedEpi e{};
            e.mode = FUSED_SWIGLU; e.ss_in = g.ssB.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = nt_b; e.H_in = c.hidden; e.eps = c.eps;
            e.act = g.act.p; e.act_lo = lo_act;
            timed(M, 2 * c.inter, c.hidden, [&] { wide_gemm(w.gate_up_il, g.xn.p, c.hidden, 0, M, 2 * c.inter, c.hidden, e); });
        }
        return k_split_resid(w.down, g.act.p, lo_act, M, c.hidden, c.inter, nullptr, true, next_norm, g.ssA.as<float>());
    }
    // The same layer for at most 128 tokens of a decode pass with every GEMM input in MFMA fragment order (k_gemm_frag.hip; g.xnp /
    // g.attnp / g.actp instead of g.xn / g.attn / g.act): five launches, none of which stages its X tile through LDS.
    static bool frag_layer_covered(const StackW& s) {
        const StackCfg& c = s.cfg;
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        for (const LayerW& w : s.layers)
            if (!w.qkv.packed || !w.o.packed || !w.gate_up_il.packed || !w.down.packed || w.ls_attn || w.ls_mlp) return false;
        return gemm_frag_covers(nqkv, c.hidden, FUSED_PLAIN) && gemm_frag_covers(c.hidden, nq, FUSED_RESID) && gemm_frag_covers(2 * c.inter, c.hidden, FUSED_SWIGLU) &&
               gemm_frag_covers(c.hidden, c.inter, FUSED_RESID);
    }
    int layer_frag(const StackW& s, int l, int M, const KVView& kv, int kv_layer, const float* next_norm, int nt_in, const float* qkv_rows = nullptr) {
        const StackCfg& c = s.cfg;
        const LayerW& w = s.layers[(size_t)l];
        const int nq = c.nh * c.d, nqkv = (c.nh + 2 * c.nkv) * c.d;
        const int prow = M <= 64 ? 64 : 128;  // packed rows of the fragment-order buffers
        QTTS_CHECK(M <= 128 && uni_len > 0 && g.xnp.p && g.attnp.p && g.actp.p && g.ss_ld >= prow, ERR_STATE, "layer_frag: a decode pass of at most 128 tokens at one position");
        if (!qkv_rows) {
            FusedEpi e{};
            e.mode = FUSED_PLAIN; e.ss_in = g.ssA.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = nt_in; e.H_in = c.hidden; e.eps = c.eps;
            e.bias = w.qkv_bias; e.out_f32 = g.part.as<float>(); e.ld_out = nqkv;
            timed(M, nqkv, c.hidden, [&] { QTTS_CHECK(launch_gemm_frag(st, w.qkv, g.xnp.p, nqkv, c.hidden, e, prow), ERR_SHAPE, "layer_frag: q|k|v shape"); });
        }
        QkvPostArgs qa{};
        qa.part = qkv_rows ? qkv_rows : g.part.as<float>(); qa.KS = 1; qa.ks_stride = 0; qa.bias = nullptr;
        qa.q_norm = w.q_norm; qa.k_norm = w.k_norm; qa.eps = c.eps; qa.cos = s.cos.as<float>(); qa.sin = s.sin.as<float>();
        qa.rope_mode = 0; qa.toks = nullptr; qa.M = M; qa.nh = c.nh; qa.nkv = c.nkv; qa.d = c.d; qa.layer = kv_layer;
        qa.q_out = g.q.as<float>(); qa.uni_len = uni_len; qa.out_packed = nq; qa.small = tune_shared ? 1 : 0;
        launch_attention_decode(st, dt, qa, kv, (float)std::pow((double)c.d, -0.5), g.attnp.p, 0);
        {
            FusedEpi e{};
            e.mode = FUSED_RESID; e.ss_ld = g.ss_ld; e.bias = w.o_bias; e.resid = g.x.as<float>(); e.x_out = g.x.as<float>();
            e.xb_out = g.xnp.p; e.nw = w.ln2; e.ss_out = g.ssB.as<float>();
            timed(M, c.hidden, nq, [&] { QTTS_CHECK(launch_gemm_frag(st, w.o, g.attnp.p, c.hidden, nq, e, prow), ERR_SHAPE, "layer_frag: o shape"); });
        }
        {
            FusedEpi e{};
            e.mode = FUSED_SWIGLU; e.ss_in = g.ssB.as<float>(); e.ss_ld = g.ss_ld; e.ntiles_in = c.hidden / 16; e.H_in = c.hidden; e.eps = c.eps;
            e.act = g.actp.p;
            timed(M, 2 * c.inter, c.hidden, [&] { QTTS_CHECK(launch_gemm_frag(st, w.gate_up_il, g.xnp.p, 2 * c.inter, c.hidden, e, prow), ERR_SHAPE, "layer_frag: gate|up shape"); });
        }
        {
            FusedEpi e{};
            e.mode = FUSED_RESID; e.ss_ld = g.ss_ld; e.resid = g.x.as<float>(); e.x_out = g.x.as<float>();
            e.xb_out = g.xnp.p; e.nw = next_norm; e.ss_out = g.ssA.as<float>();
            timed(M, c.hidden, c.inter, [&] { QTTS_CHECK(launch_gemm_frag(st, w.down, g.actp.p, c.hidden, c.inter, e, prow), ERR_SHAPE, "layer_frag: down shape"); });
        }
        return c.hidden / 16;
    }
    // whole stack; afterwards xn = T(x * final_norm) and ssA holds the returned number of tiles
    // qkv_rows0: q|k|v rows of the first layer when they are already there (see layer_fused)
    int stack_fused(const StackW& s, int M, const Tok* toks, const KVView& kv, int nt_in, bool decode, const float* qkv_rows0 = nullptr) {
        int nt = nt_in;
        for (int l = 0; l < s.cfg.layers; ++l) {
            const bool last = l + 1 == s.cfg.layers;
            nt = layer_fused(s, l, M, toks, kv, l, last ? s.final_norm : s.layers[(size_t)l + 1].ln1, nt, decode, l == 0 ? qkv_rows0 : nullptr);
        }
        return nt;
    }

    // kv_per_layer: the KV view holds every layer (decode caches); otherwise one layer's worth is reused
    // final_xn_lo: remainder-plane offset of final_xn (0 = none)
    void stack(const StackW& s, int M, const Tok* toks, const KVView& kv, void* final_xn, float* final_xn_f32, bool kv_per_layer = true,
               bool decode = false, long final_xn_lo = 0) {
        xn_tiles = 0;  // the caller's xn is normalised
        for (int l = 0; l < s.cfg.layers; ++l) {
            const bool last = l + 1 == s.cfg.layers;
            layer(s, l, M, toks, kv, kv_per_layer ? l : 0, last ? s.final_norm : s.layers[(size_t)l + 1].ln1, last ? final_xn : g.xn.p,
                  last ? final_xn_f32 : nullptr, decode, last ? final_xn_lo : lo_xn);
        }
    }
};

}  // namespace qtts
