// Attention for tokens in flight over a paged KV cache (gfx950).
//
// qkv_post: split-K combine of the fused q|k|v projection, per-head RMSNorm on q and k, RoPE, KV append
//   (qwen_tts/src/nn/attention/unified.rs:193-216, nn/rope/mod.rs:46-62, nn/kv_cache.rs:38-42).
// attention: GQA softmax(q k^T * scale) v for one token against its row's cached keys
//   (nn/attention/mod.rs:359-472; the additive 0/-inf masks become the key range in Tok).
// Both are HBM-bound: K/V are read once, straight to registers with 16-byte loads (8 head dims per
// lane, D/8 lanes per key), online softmax in f32, one block per (token, kv head).
#include "attn_common.hpp"

#include <cstdlib>

namespace qtts {

template <typename T>
__global__ __launch_bounds__(64) void qkv_post_kernel(QkvPostArgs a, KVView kv) {
    const int m = blockIdx.x, hi = blockIdx.y, lane = threadIdx.x;
    const int d = a.d, half = d >> 1;
    const int ncol = (a.nh + 2 * a.nkv) * d;
    const Tok t = a.toks[m];
    const bool act = lane < half;
    const int i = act ? lane : 0;
    const long col = (long)hi * d + i;
    float x1, x2;
    slab_sum_pair(a.part + (long)m * ncol + col, half, a.KS, a.ks_stride, x1, x2);
    if (a.bias) {
        x1 += a.bias[col];
        x2 += a.bias[col + half];
    }
    const bool is_q = hi < a.nh, is_k = !is_q && hi < a.nh + a.nkv;
    if (is_q || is_k) {
        const float* nw = is_q ? a.q_norm : a.k_norm;
        if (nw) {
            const float ss = wave_sum(act ? x1 * x1 + x2 * x2 : 0.f);
            const float mm = sqrtf(ss / (float)d + a.eps);
            x1 = x1 / mm * nw[i];
            x2 = x2 / mm * nw[i + half];
        }
        const float c = a.cos[(long)t.pos * half + i], s = a.sin[(long)t.pos * half + i];
        const float y1 = x1 * c - x2 * s;
        const float y2 = x1 * s + x2 * c;
        x1 = y1;
        x2 = y2;
    }
    if (!act) return;
    if (is_q) {
        float* q = a.q_out + (long)m * a.nh * d + (long)hi * d;
        q[i] = x1;
        q[i + half] = x2;
    } else {
        const int g = is_k ? hi - a.nh : hi - a.nh - a.nkv;
        T* dst = kv_ptr<T>(kv, a.layer, t.row, t.pos, is_k ? 0 : 1, g);
        dst[i] = from_f32<T>(x1);
        dst[i + half] = from_f32<T>(x2);
    }
}

void launch_qkv_post(hipStream_t st, int dtype, const QkvPostArgs& a, const KVView& kv) {
    QTTS_CHECK(a.d == 64 || a.d == 128, ERR_SHAPE, "qkv_post: head_dim must be 64 or 128");
    QTTS_CHECK(a.M >= 1 && a.nh >= 1 && a.nkv >= 1 && kv.d == a.d && kv.nkv == a.nkv, ERR_SHAPE, "qkv_post: bad shape");
    dim3 grid(a.M, a.nh + 2 * a.nkv);
    if (dtype == BF16) hipLaunchKernelGGL(qkv_post_kernel<bf16_t>, grid, dim3(64), 0, st, a, kv);
    else hipLaunchKernelGGL(qkv_post_kernel<float>, grid, dim3(64), 0, st, a, kv);
}

// ------------------------------------------------------------------------------------------------
template <typename T, int D, int REP>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const Tok* __restrict__ toks, KVView kv,
                                                        int layer, int nh, float scale, T* out, long out_lo) {
    constexpr int LPK = D / 8;     // lanes per key (8 head dims each)
    constexpr int KPW = 64 / LPK;  // keys per wave per iteration
    __shared__ float sm[4][REP][LPK][10];
    const int m = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gi = lane / LPK, li = lane % LPK;
    const Tok t = toks[m];

    float qv[REP][8], o[REP][8], mx[REP], l[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const float* qp = q + ((long)m * nh + (long)g * REP + h) * D + li * 8;
        load8(qp, qv[h]);
        mx[h] = -1e30f;
        l[h] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[h][e] = 0.f;
    }

    for (int base = wave * KPW; base < t.ctx_len; base += 4 * KPW) {
        const int key = base + gi;
        const bool valid = key < t.ctx_len;
        const int pos = t.ctx_start + (valid ? key : 0);
        const T* kp = kv_ptr<T>(kv, layer, t.row, pos, 0, g) + li * 8;
        const T* vp = kv_ptr<T>(kv, layer, t.row, pos, 1, g) + li * 8;
        float kf[8], vf[8];
        load8(kp, kf);
        load8(vp, vf);
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qv[h][e] * kf[e];
#pragma unroll
            for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            s = valid ? s * scale : -INFINITY;
            const float mn = fmaxf(mx[h], s);
            const float alpha = expf(mx[h] - mn);
            const float p = expf(s - mn);
            l[h] = l[h] * alpha + p;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[h][e] = o[h][e] * alpha + p * vf[e];
            mx[h] = mn;
        }
    }
    // merge the KPW key groups of this wave
#pragma unroll
    for (int h = 0; h < REP; ++h) {
#pragma unroll
        for (int off = LPK; off < 64; off <<= 1) {
            const float mo = __shfl_xor(mx[h], off, 64), lo = __shfl_xor(l[h], off, 64);
            const float mn = fmaxf(mx[h], mo);
            const float a1 = expf(mx[h] - mn), a2 = expf(mo - mn);
            l[h] = l[h] * a1 + lo * a2;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[h][e] = o[h][e] * a1 + __shfl_xor(o[h][e], off, 64) * a2;
            mx[h] = mn;
        }
        if (gi == 0) {
            sm[wave][h][li][0] = mx[h];
            sm[wave][h][li][1] = l[h];
#pragma unroll
            for (int e = 0; e < 8; ++e) sm[wave][h][li][2 + e] = o[h][e];
        }
    }
    __syncthreads();
    if (threadIdx.x < REP * LPK) {
        const int h = threadIdx.x / LPK, j = threadIdx.x % LPK;
        float mn = sm[0][h][j][0];
        for (int w = 1; w < 4; ++w) mn = fmaxf(mn, sm[w][h][j][0]);
        float lt = 0.f, ot[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int w = 0; w < 4; ++w) {
            const float aw = expf(sm[w][h][j][0] - mn);
            lt += sm[w][h][j][1] * aw;
#pragma unroll
            for (int e = 0; e < 8; ++e) ot[e] += sm[w][h][j][2 + e] * aw;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ot[e] = ot[e] / lt;
        store8_split<T>(out + ((long)m * nh + (long)g * REP + h) * D + j * 8, out_lo, ot);
    }
}

// sum over the LPK lanes that share a key, left in all of them (butterfly LPK/2 .. 1)
template <int LPK>
__device__ __forceinline__ float key_group_sum(float s) {
    if constexpr (LPK == 16) {
        return row16_sum(s);
    } else {
#pragma unroll
        for (int off = LPK / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
        return s;
    }
}

// ------------------------------------------------------------------------------------------------
// Decode-step variant: the token's own q/k/v post-processing (split-K combine, bias, per-head RMSNorm, RoPE, KV
// append) happens inside the attention block, so one launch replaces qkv_post + attention.  Valid only when a
// token's newest key is its own position and no other token of the launch attends to it (decode steps, the code
// predictor's single-token passes); prefill chunks use the two-kernel path.
template <typename T, int D, int REP, int NG, int NW>
__global__ __launch_bounds__(NW * 64) void attention_decode_kernel(QkvPostArgs a, KVView kv, float scale, T* out, long out_lo) {
    constexpr int LPK = D / 8, KPW = 64 / LPK, HALF = D / 2;
    constexpr int NP = NW * KPW;  // partial results of a block: one per (wave, lane group)
    static_assert(REP * NP <= NW * 64, "the merge needs one thread per (head, partial)");
    __shared__ float po[NP][REP][D];         // unnormalised weighted sums of V
    __shared__ float pm[NP][REP], pl[NP][REP], pa[NP][REP], pL[REP];
    __shared__ float qs[REP][D];
    __shared__ float kown[D], vown[D];
    QTTS_TS(0);
    const int m = blockIdx.x, g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // a launch whose tokens all sit at one position (the code predictor's passes) describes them in the arguments: one
    // dependent load less in front of the K/V rows
    const Tok t = a.uni_len > 0 ? Tok{m, a.uni_len - 1, 0, a.uni_len} : a.toks[m];
    const int ncol = (a.nh + 2 * a.nkv) * D;
    // The key loop is software-pipelined by hand: a fetch covers NG groups of KPW keys per wave, and the K/V rows (with the
    // page-table entries in front of them) of the next fetch are in flight while the current one goes through the online softmax.
    // One group per fetch keeps the block at 120 VGPRs (four blocks per CU instead of three at two groups): with several
    // batches in flight the extra resident blocks are worth more than the deeper prefetch.
    // The first fetch is issued before phase 1, so that the cached rows travel while the token's own q/k/v are put together;
    // the newest key is not in the cache yet and is patched in from LDS (own_key) after the barrier.
    const int gi = lane / LPK, li = lane % LPK;
    auto fetch = [&](int base, float(&kf)[NG][8], float(&vf)[NG][8]) {
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const int key = base + n * KPW + gi;
            const bool cached = key < t.ctx_len - 1;
            const int pos = t.ctx_start + (cached ? key : 0);
            load8(kv_ptr<T>(kv, a.layer, t.row, pos, 0, g) + li * 8, kf[n]);
            load8(kv_ptr<T>(kv, a.layer, t.row, pos, 1, g) + li * 8, vf[n]);
        }
    };
    auto own_key = [&](int base, float(&kf)[NG][8], float(&vf)[NG][8]) {
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            if (base + n * KPW + gi == t.ctx_len - 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { kf[n][e] = kown[li * 8 + e]; vf[n][e] = vown[li * 8 + e]; }
            }
        }
    };
    float kA[NG][8], vA[NG][8], kB[NG][8], vB[NG][8];
    const int base0 = wave * NG * KPW;
    if (base0 < t.ctx_len) fetch(base0, kA, vA);
    // ---- phase 1: k, v and the REP query heads of this kv group (one wave per head, lane = rotation pair)
    for (int hh = wave; hh < 2 + REP; hh += NW) {
        const bool is_k = hh == 0, is_v = hh == 1;
        const int head = is_k ? a.nh + g : (is_v ? a.nh + a.nkv + g : g * REP + (hh - 2));
        const bool act = lane < HALF;
        const int i = act ? lane : 0;
        const long col = (long)head * D + i;
        float x1, x2;
        slab_sum_pair(a.part + (long)m * ncol + col, HALF, a.KS, a.ks_stride, x1, x2);
        if (a.bias) {
            x1 += a.bias[col];
            x2 += a.bias[col + HALF];
        }
        if (!is_v) {
            const float* nw = is_k ? a.k_norm : a.q_norm;
            if (nw) {
                const float ss = wave_sum(act ? x1 * x1 + x2 * x2 : 0.f);
                const float mm = sqrtf(ss / (float)D + a.eps);
                x1 = x1 / mm * nw[i];
                x2 = x2 / mm * nw[i + HALF];
            }
            const float c = a.cos[(long)t.pos * HALF + i], s = a.sin[(long)t.pos * HALF + i];
            const float y1 = x1 * c - x2 * s, y2 = x1 * s + x2 * c;
            x1 = y1;
            x2 = y2;
        }
        if (act) {
            if (is_k || is_v) {
                T* dst = kv_ptr<T>(kv, a.layer, t.row, t.pos, is_k ? 0 : 1, g);
                const T r1 = from_f32<T>(x1), r2 = from_f32<T>(x2);
                dst[i] = r1;
                dst[i + HALF] = r2;
                float* own = is_k ? kown : vown;
                own[i] = to_f32<T>(r1);   // the value later steps will read back from the cache
                own[i + HALF] = to_f32<T>(r2);
            } else {
                qs[hh - 2][i] = x1;
                qs[hh - 2][i + HALF] = x2;
            }
        }
    }
    QTTS_TS(1);
    __syncthreads();
    QTTS_TS(2);
    // ---- phase 2: online-softmax attention; the newest key comes from LDS, the rest from the cache
    float qv[REP][8], o[REP][8], mx[REP], l[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { qv[h][e] = qs[h][li * 8 + e]; o[h][e] = 0.f; }
        mx[h] = -1e30f;
        l[h] = 0.f;
    }
    auto accumulate = [&](int base, const float(&kf)[NG][8], const float(&vf)[NG][8]) {
#pragma unroll
        for (int n = 0; n < NG; ++n) {
            const bool valid = base + n * KPW + gi < t.ctx_len;
            float vz[8];  // a padding lane's row is whatever the cache holds: it must not meet its zero weight as inf or NaN
#pragma unroll
            for (int e = 0; e < 8; ++e) vz[e] = valid ? vf[n][e] : 0.f;
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s += qv[h][e] * kf[n][e];
                s = key_group_sum<LPK>(s);
                s = valid ? s * scale : -INFINITY;
                // online softmax step: of exp(mx - mn) and exp(s - mn) one is exp(0) = 1, so one exponential serves both
                const bool up = s > mx[h];
                const float ex = expf(up ? mx[h] - s : s - mx[h]);
                const float alpha = up ? ex : 1.0f, p = up ? 1.0f : ex;
                l[h] = l[h] * alpha + p;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[h][e] = o[h][e] * alpha + p * vz[e];
                mx[h] = up ? s : mx[h];
            }
        }
    };
    {
        constexpr int STEP = NW * NG * KPW;
        int base = base0;
        while (base < t.ctx_len) {
            if (base + STEP < t.ctx_len) fetch(base + STEP, kB, vB);
            own_key(base, kA, vA);
            accumulate(base, kA, vA);
            base += STEP;
            if (base >= t.ctx_len) break;
            if (base + STEP < t.ctx_len) fetch(base + STEP, kA, vA);
            own_key(base, kB, vB);
            accumulate(base, kB, vB);
            base += STEP;
        }
    }
    QTTS_TS(3);
    // ---- merge of the NP partial softmaxes (running maximum m, sum l, weighted V sum o): every lane group hands its own over
    // through LDS; REP * NP threads turn the maxima into weights a = exp(m - M) and the total L = sum a l; then one thread per
    // output element adds the weighted partials in partial order.
    {
        const int part = wave * KPW + gi;
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            *reinterpret_cast<float4*>(&po[part][h][li * 8]) = make_float4(o[h][0], o[h][1], o[h][2], o[h][3]);
            *reinterpret_cast<float4*>(&po[part][h][li * 8 + 4]) = make_float4(o[h][4], o[h][5], o[h][6], o[h][7]);
            if (li == 0) {
                pm[part][h] = mx[h];
                pl[part][h] = l[h];
            }
        }
    }
    QTTS_TS(4);
    __syncthreads();
    QTTS_TS(5);
    if (threadIdx.x < REP * NP) {
        const int h = threadIdx.x / NP, part = threadIdx.x % NP;
        const float m = pm[part][h];
        float M, L;
        if constexpr (NP == 16) {  // the partials of a head sit in one 16-lane row
            M = m;
            M = fmaxf(M, dpp_move<DPP_ROW_ROR + 8>(M));
            M = fmaxf(M, dpp_move<DPP_ROW_ROR + 4>(M));
            M = fmaxf(M, dpp_move<DPP_ROW_ROR + 2>(M));
            M = fmaxf(M, dpp_move<DPP_ROW_ROR + 1>(M));
            const float aw = expf(m - M);
            pa[part][h] = aw;
            L = row16_sum(aw * pl[part][h]);
        } else if constexpr (NP == 32 || NP == 64) {  // the partials of a head sit in NP consecutive lanes of one wave
            M = m;
#pragma unroll
            for (int off = NP / 2; off > 0; off >>= 1) M = fmaxf(M, __shfl_xor(M, off, 64));
            const float aw = expf(m - M);
            pa[part][h] = aw;
            L = aw * pl[part][h];
#pragma unroll
            for (int off = NP / 2; off > 0; off >>= 1) L += __shfl_xor(L, off, 64);
        } else {
            M = pm[0][h];
            for (int q = 1; q < NP; ++q) M = fmaxf(M, pm[q][h]);
            pa[part][h] = expf(m - M);
            L = 0.f;
            for (int q = 0; q < NP; ++q) L += expf(pm[q][h] - M) * pl[q][h];
        }
        if (part == 0) pL[h] = L;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < REP * D; idx += NW * 64) {
        const int h = idx / D, dd = idx % D;
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) acc += pa[q][h] * po[q][h][dd];
        if (a.out_packed > 0) out[packed_elem(m, (g * REP + h) * D + dd, a.out_packed)] = from_f32<T>(acc / pL[h]);
        else store_split<T>(out + ((long)m * a.nh + (long)g * REP + h) * D + dd, out_lo, acc / pL[h]);
    }
    QTTS_TS(6);
}

// Decode-step variant for short contexts (the code predictor's passes: every token at one position, at most 16 keys, head_dim
// 128, bf16, finished q|k|v rows): ONE WAVE per (token, kv head) does what a block of four does above - the token's k, v and REP
// query heads (RMSNorm, RoPE, KV append), then softmax(q k^T) v over the cached keys plus its own.  Four key groups of 16 lanes
// take keys gi, gi + 4, gi + 8, gi + 12, so the whole context is one fetch, issued before anything else; the softmax is the plain
// two-pass one (all scores, maximum, weights), merged across the key groups with the permlane swaps.  No block barrier: a wave
// passes its own rows from the rotation-pair layout of phase 1 to the eight-dims-per-lane layout of phase 2 through its slice of
// LDS.  A quarter of the waves and blocks of the general kernel for these launches, which is what counts with several decode
// streams in flight.
// Measured on the 1.7B shapes (70 such launches per frame): three sessions in flight +2.3 % audio-s/s; one stream alone 0.12 ms per
// frame slower than the general kernel (0.04 ms with two waves per pair), so only runs with several streams select it.
template <int REP>
__global__ __launch_bounds__(256) void attention_small_kernel(QkvPostArgs a, KVView kv, float scale, bf16_t* out) {
    using T = bf16_t;
    constexpr int D = 128, HALF = 64, NK = 4;
    __shared__ float shm[4][2 + REP][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.M * a.nkv) return;
    float (*sh)[D] = shm[wave];
    const int m = pair / a.nkv, g = pair - m * a.nkv;
    const int ctx = a.uni_len, pos = ctx - 1;
    const int gi = lane >> 4, li = lane & 15;
    const int ncol = (a.nh + 2 * a.nkv) * D;
    // the cached keys and values (a lane group's own key slot is patched in from LDS later; slots past the context are masked)
    float kf[NK][8], vf[NK][8];
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        const int key = n * 4 + gi;
        const int p = key < pos ? key : 0;
        load8(kv_ptr<T>(kv, a.layer, m, p, 0, g) + li * 8, kf[n]);
        load8(kv_ptr<T>(kv, a.layer, m, p, 1, g) + li * 8, vf[n]);
    }
    // ---- phase 1: lane = rotation pair (i, i + 64) of k, v and the REP query heads
    const float* prow = a.part + (long)m * ncol;
    float x1[2 + REP], x2[2 + REP];
#pragma unroll
    for (int hh = 0; hh < 2 + REP; ++hh) {
        const int head = hh == 0 ? a.nh + g : (hh == 1 ? a.nh + a.nkv + g : g * REP + (hh - 2));
        x1[hh] = prow[head * D + lane];
        x2[hh] = prow[head * D + lane + HALF];
    }
    const float cs = a.cos[(long)pos * HALF + lane], sn = a.sin[(long)pos * HALF + lane];
    const float kw1 = a.k_norm ? a.k_norm[lane] : 1.f, kw2 = a.k_norm ? a.k_norm[lane + HALF] : 1.f;
    const float qw1 = a.q_norm ? a.q_norm[lane] : 1.f, qw2 = a.q_norm ? a.q_norm[lane + HALF] : 1.f;
#pragma unroll
    for (int hh = 0; hh < 2 + REP; ++hh) {
        float y1 = x1[hh], y2 = x2[hh];
        if (hh != 1) {
            if (hh == 0 ? a.k_norm != nullptr : a.q_norm != nullptr) {
                const float ss = wave_sum(y1 * y1 + y2 * y2);
                const float mm = sqrtf(ss / (float)D + a.eps);
                y1 = y1 / mm * (hh == 0 ? kw1 : qw1);
                y2 = y2 / mm * (hh == 0 ? kw2 : qw2);
            }
            const float r1 = y1 * cs - y2 * sn, r2 = y1 * sn + y2 * cs;
            y1 = r1;
            y2 = r2;
        }
        if (hh < 2) {
            T* dst = kv_ptr<T>(kv, a.layer, m, pos, hh, g);
            const T b1 = from_f32<T>(y1), b2 = from_f32<T>(y2);
            dst[lane] = b1;
            dst[lane + HALF] = b2;
            y1 = to_f32<T>(b1);  // the value later passes will read back from the cache
            y2 = to_f32<T>(b2);
        }
        sh[hh][lane] = y1;
        sh[hh][lane + HALF] = y2;
    }
    // the wave's own LDS writes, then its reads in the other layout: LDS serves one wave's accesses in order
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- phase 2: lane = dims li * 8 .. + 7 of the keys of group gi
    float qv[REP][8], ko[8], vo[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ko[e] = sh[0][li * 8 + e]; vo[e] = sh[1][li * 8 + e]; }
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
        for (int e = 0; e < 8; ++e) qv[h][e] = sh[2 + h][li * 8 + e];
    float sc[NK][REP], mx[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) mx[h] = -INFINITY;
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        const int key = n * 4 + gi;
        const bool own = key == pos, valid = key < ctx;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            kf[n][e] = own ? ko[e] : kf[n][e];
            vf[n][e] = valid ? (own ? vo[e] : vf[n][e]) : 0.f;  // a padding slot's row is whatever the cache holds: keep inf / NaN out
        }
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) s += qv[h][e] * kf[n][e];
            s = row16_sum(s);
            sc[n][h] = valid ? s * scale : -INFINITY;
            mx[h] = fmaxf(mx[h], sc[n][h]);
        }
    }
    const auto fmx = [](float x, float y) { return fmaxf(x, y); };
    const auto add = [](float x, float y) { return x + y; };
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        mx[h] = fold_rows(fold_halves(mx[h], fmx), fmx);  // key 0 is always valid: the maximum is finite
        float l = 0.f, o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = 0.f;
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const float p = expf(sc[n][h] - mx[h]);
            l += p;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += p * vf[n][e];
        }
        l = fold_rows(fold_halves(l, add), add);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = fold_rows(fold_halves(o[e], add), add);
        if (gi == 0) {
            const float inv = 1.0f / l;
            u16 hb[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) hb[e] = f32_to_bf16_bits(o[e] * inv);
            const int col = (g * REP + h) * D + li * 8;
            T* dst = a.out_packed > 0 ? out + packed_elem(m, col, a.out_packed) : out + (long)m * a.nh * D + col;
            *reinterpret_cast<uint4*>(dst) = make_uint4((unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16),
                                                        (unsigned)hb[4] | ((unsigned)hb[5] << 16), (unsigned)hb[6] | ((unsigned)hb[7] << 16));
        }
    }
}

// QTTS_ATTN_SMALL=0: short-context decode launches through the general kernel
static bool attn_small_enabled() {
    static const bool on = !(getenv("QTTS_ATTN_SMALL") && atoi(getenv("QTTS_ATTN_SMALL")) == 0);
    return on;
}

// QTTS_ATTN_WAVES=4 / 8: waves per block of the launches over growing contexts, whatever else shares the GPU
static int attn_long_waves(bool shared) {
    static const int nw = getenv("QTTS_ATTN_WAVES") ? atoi(getenv("QTTS_ATTN_WAVES")) : 0;
    return nw == 4 || nw == 8 ? nw : (shared ? 4 : 8);
}

template <typename T, int D, int NG>
static void launch_attn_decode_ng(hipStream_t st, const QkvPostArgs& a, const KVView& kv, float scale, T* out, long out_lo) {
    const int rep = a.nh / kv.nkv;
    dim3 grid(a.M, kv.nkv), block(256);
    // A block walks its token's keys NW * 64 / (D / 8) at a time with two fetches in flight per wave, and the grid (tokens x
    // kv heads) is only two blocks per CU: with contexts of hundreds of keys the walk is bound by the bytes in flight.  A
    // stream that has the GPU to itself takes eight waves per block for such launches (frame of 64 rows alone 4.48 -> 4.42 ms);
    // with three sessions in flight the other sessions' kernels fill the CUs and four waves measured 3.7 % more audio-s/s.
    // The code predictor's passes (at most 17 keys, one position for every token) keep four.
    if constexpr (D == 128 && NG == 1) {
        if (rep == 2 && a.uni_len == 0 && attn_long_waves(a.small != 0) == 8) {
            hipLaunchKernelGGL((attention_decode_kernel<T, D, 2, NG, 8>), grid, dim3(512), 0, st, a, kv, scale, out, out_lo);
            return;
        }
    }
    switch (rep) {
        case 1: hipLaunchKernelGGL((attention_decode_kernel<T, D, 1, NG, 4>), grid, block, 0, st, a, kv, scale, out, out_lo); break;
        case 2: hipLaunchKernelGGL((attention_decode_kernel<T, D, 2, NG, 4>), grid, block, 0, st, a, kv, scale, out, out_lo); break;
        case 4: hipLaunchKernelGGL((attention_decode_kernel<T, D, 4, NG, 4>), grid, block, 0, st, a, kv, scale, out, out_lo); break;
        case 8: hipLaunchKernelGGL((attention_decode_kernel<T, D, 8, NG, 4>), grid, block, 0, st, a, kv, scale, out, out_lo); break;
        default: throw Error(ERR_SHAPE, "attention: heads per kv head must be 1, 2, 4 or 8");
    }
}
template <typename T, int D>
static void launch_attn_decode_rep(hipStream_t st, const QkvPostArgs& a, const KVView& kv, float scale, T* out, long out_lo) {
    static const int ng = getenv("QTTS_ATTN_NG") ? atoi(getenv("QTTS_ATTN_NG")) : 1;
    if (ng == 4 && a.nh / kv.nkv <= 2) launch_attn_decode_ng<T, D, 4>(st, a, kv, scale, out, out_lo);
    else if (ng >= 2 && a.nh / kv.nkv <= 2) launch_attn_decode_ng<T, D, 2>(st, a, kv, scale, out, out_lo);
    else launch_attn_decode_ng<T, D, 1>(st, a, kv, scale, out, out_lo);
}

void launch_attention_decode(hipStream_t st, int dtype, const QkvPostArgs& a, const KVView& kv, float scale, void* out, long out_lo) {
    QTTS_CHECK(a.d == 64 || a.d == 128, ERR_SHAPE, "attention: head_dim must be 64 or 128");
    QTTS_CHECK(a.M >= 1 && a.nh >= 1 && a.nkv >= 1 && a.nh % a.nkv == 0 && kv.d == a.d && kv.nkv == a.nkv, ERR_SHAPE, "attention: bad shape");
    QTTS_CHECK(a.out_packed == 0 || (dtype == BF16 && out_lo == 0 && a.M <= 128 && a.out_packed == a.nh * a.d && a.out_packed % 64 == 0), ERR_BAD_ARG,
               "attention: a fragment-order output needs bf16, one plane, at most 128 tokens and a width of nh * d");
    if (dtype == BF16 && a.d == 128 && a.small && a.uni_len >= 1 && a.uni_len <= 16 && a.KS == 1 && !a.bias && out_lo == 0 && attn_small_enabled() &&
        (a.nh == kv.nkv || a.nh == 2 * kv.nkv)) {
        const dim3 grid((unsigned)((a.M * kv.nkv + 3) / 4));
        if (a.nh == kv.nkv) hipLaunchKernelGGL(attention_small_kernel<1>, grid, dim3(256), 0, st, a, kv, scale, (bf16_t*)out);
        else hipLaunchKernelGGL(attention_small_kernel<2>, grid, dim3(256), 0, st, a, kv, scale, (bf16_t*)out);
        return;
    }
    if (dtype == BF16) {
        if (a.d == 128) launch_attn_decode_rep<bf16_t, 128>(st, a, kv, scale, (bf16_t*)out, out_lo);
        else launch_attn_decode_rep<bf16_t, 64>(st, a, kv, scale, (bf16_t*)out, out_lo);
    } else {
        QTTS_CHECK(out_lo == 0, ERR_BAD_ARG, "attention: a remainder plane exists for bf16 activations only");
        if (a.d == 128) launch_attn_decode_rep<float, 128>(st, a, kv, scale, (float*)out, 0);
        else launch_attn_decode_rep<float, 64>(st, a, kv, scale, (float*)out, 0);
    }
}

template <typename T, int D>
static void launch_attn_rep(hipStream_t st, const float* q, const Tok* toks, const KVView& kv, int layer, int M, int nh,
                            float scale, T* out, long out_lo) {
    const int rep = nh / kv.nkv;
    dim3 grid(M, kv.nkv), block(256);
    switch (rep) {
        case 1: hipLaunchKernelGGL((attention_kernel<T, D, 1>), grid, block, 0, st, q, toks, kv, layer, nh, scale, out, out_lo); break;
        case 2: hipLaunchKernelGGL((attention_kernel<T, D, 2>), grid, block, 0, st, q, toks, kv, layer, nh, scale, out, out_lo); break;
        case 4: hipLaunchKernelGGL((attention_kernel<T, D, 4>), grid, block, 0, st, q, toks, kv, layer, nh, scale, out, out_lo); break;
        case 8: hipLaunchKernelGGL((attention_kernel<T, D, 8>), grid, block, 0, st, q, toks, kv, layer, nh, scale, out, out_lo); break;
        default: throw Error(ERR_SHAPE, "attention: heads per kv head must be 1, 2, 4 or 8");
    }
}

void launch_attention(hipStream_t st, int dtype, const float* q, const Tok* toks, const KVView& kv, int layer, int M, int nh,
                      float scale, void* out, long out_lo) {
    QTTS_CHECK(kv.d == 64 || kv.d == 128, ERR_SHAPE, "attention: head_dim must be 64 or 128");
    QTTS_CHECK(M >= 1 && nh % kv.nkv == 0, ERR_SHAPE, "attention: bad head counts");
    if (dtype == BF16) {
        if (kv.d == 128) launch_attn_rep<bf16_t, 128>(st, q, toks, kv, layer, M, nh, scale, (bf16_t*)out, out_lo);
        else launch_attn_rep<bf16_t, 64>(st, q, toks, kv, layer, M, nh, scale, (bf16_t*)out, out_lo);
    } else {
        QTTS_CHECK(out_lo == 0, ERR_BAD_ARG, "attention: a remainder plane exists for bf16 activations only");
        if (kv.d == 128) launch_attn_rep<float, 128>(st, q, toks, kv, layer, M, nh, scale, (float*)out, 0);
        else launch_attn_rep<float, 64>(st, q, toks, kv, layer, M, nh, scale, (float*)out, 0);
    }
}

}  // namespace qtts
