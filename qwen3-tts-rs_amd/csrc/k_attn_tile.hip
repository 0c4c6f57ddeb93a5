// Tiled attention on the matrix cores (gfx950) for launches whose tokens come in runs: prefill chunks of the talker
// (qwen_tts/src/nn/attention/mod.rs:359-472, causal / sliding-window masks :199-266), the codec transformer's banded
// attention (audio/tokenizer/v2/transformer.rs:357-388, 72 keys per query) and the codec encoder's causal transformer.
//
// A block owns up to 64 consecutive tokens of one cache row (AttnTile) and one kv head; its four waves take 16 queries each
// for every query head of the kv group.  The keys and values the tile's queries can see are walked in blocks of 32
// positions: each block is fetched from the paged cache ONCE per tile into registers (the fetch of block i + 1 travels while
// block i is multiplied), staged in LDS and shared by the waves - instead of once per token as in the one-token kernel.
//   S^T = K Q^T   : K rows from LDS are the A operand, the wave's queries the B operand, so a lane holds the scores of four
//                   consecutive keys for ONE query (row r = lane & 15); a query's softmax statistics live in the four lanes
//                   r, r + 16, r + 32, r + 48 and are combined with two permlane swaps.
//   O^T = V^T P^T : the score registers are, as they stand, the B operand of v_mfma_f32_16x16x16_bf16; V^T comes out of the
//                   row-major LDS image through the transposing read ds_read_b64_tr_b16.
// bf16 caches: queries and probabilities enter as a bf16 value plus a bf16 remainder (two MFMAs), so that only K and V
// carry bf16 rounding, as in the one-token kernel.  f32 caches use the exact v_mfma_f32_16x16x4_f32.
// Masks are the tokens' key ranges [ctx_start, ctx_start + ctx_len); scores outside are -inf before the online softmax.
#include "attn_common.hpp"

#include <algorithm>
#include <vector>

namespace qtts {

typedef short s16x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int wave_min_int(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = min(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_max_int(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, __shfl_xor(v, off, 64));
    return v;
}
// combination over the four lanes r, r + 16, r + 32, r + 48 (left in all of them)
template <typename F>
__device__ __forceinline__ float quad_rows(float v, F op) {
    v = fold_halves(v, op);
    return fold_rows(v, op);
}

template <typename T, int D, int REP>
__global__ __launch_bounds__(256) void attention_tile_kernel(const float* __restrict__ q, const Tok* __restrict__ toks,
                                                             const AttnTile* __restrict__ tiles, KVView kv, int layer, int nh,
                                                             float scale, T* out, long out_lo) {
    constexpr bool BF = sizeof(T) == 2;
    constexpr int E = 16 / (int)sizeof(T);  // elements per 16-byte piece
    constexpr int LD = D + E;               // LDS row stride: 16 bytes of padding per key row
    constexpr int PPR = D / E;              // pieces per key row
    constexpr int KB = 32;                  // key positions per block
    constexpr int NPT = KB * PPR / 256;     // pieces a thread fetches per tensor and block
    constexpr int NDT = D / 16;             // 16-column tiles of the output
    constexpr int NQ = BF ? D / 32 : D / 4; // query operand registers (groups) per head
    static_assert(NPT >= 1 && KB * PPR % 256 == 0, "tile attention: head_dim not covered");
    __shared__ __attribute__((aligned(16))) T Ks[KB * LD];
    __shared__ __attribute__((aligned(16))) T Vs[KB * LD];

    const AttnTile tl = tiles[blockIdx.x];
    const int g = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, gq = lane >> 4;
    const int qi = wave * 16 + r;  // this lane's query within the tile
    const bool vq = qi < tl.count;
    const Tok tq = toks[tl.m0 + (vq ? qi : 0)];
    // an absent query has an empty key range: every score is masked, its probabilities are zero and nothing is stored
    const int cs = vq ? tq.ctx_start : 0x7fffffff;
    const int ce = vq ? tq.ctx_start + tq.ctx_len : -0x7fffffff;
    const int wlo = wave_min_int(cs), whi = wave_max_int(ce);  // key positions any query of this wave can see

    // ---- the wave's queries as the B operand
    bf16x8_t qh[REP][BF ? NQ : 1], ql[REP][BF ? NQ : 1];
    float qf[REP][BF ? 1 : NQ];
    {
        const long qrow = (long)(tl.m0 + (vq ? qi : 0)) * nh + (long)g * REP;
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            const float* qp = q + (qrow + h) * D;
            if constexpr (BF) {
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    float f[8];
                    load8(qp + 32 * j + 8 * gq, f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const __bf16 hi = (__bf16)f[e];
                        qh[h][j][e] = hi;
                        ql[h][j][e] = (__bf16)(f[e] - (float)hi);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < NQ; ++j) qf[h][j] = qp[4 * j + gq];
            }
        }
    }

    f32x4_t acc[REP][NDT];
    float mrun[REP], lrun[REP];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        mrun[h] = -INFINITY;
        lrun[h] = 0.f;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) acc[h][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }

    // ---- staging of a key block: global -> registers -> LDS.  Positions outside [k0, k1) are not read (zeros): a stale or
    // unwritten V row must not meet its zero probability as inf or NaN.
    uint4 kreg[NPT], vreg[NPT];
    auto fetch = [&](int kb) {
#pragma unroll
        for (int n = 0; n < NPT; ++n) {
            const int idx = threadIdx.x + n * 256;
            const int key = idx / PPR, pc = idx % PPR;
            const int pos = kb + key;
            kreg[n] = make_uint4(0u, 0u, 0u, 0u);
            vreg[n] = make_uint4(0u, 0u, 0u, 0u);
            if (pos >= tl.k0 && pos < tl.k1) {
                kreg[n] = *reinterpret_cast<const uint4*>(kv_ptr<T>(kv, layer, tl.row, pos, 0, g) + pc * E);
                vreg[n] = *reinterpret_cast<const uint4*>(kv_ptr<T>(kv, layer, tl.row, pos, 1, g) + pc * E);
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int n = 0; n < NPT; ++n) {
            const int idx = threadIdx.x + n * 256;
            const int key = idx / PPR, pc = idx % PPR;
            *reinterpret_cast<uint4*>(&Ks[key * LD + pc * E]) = kreg[n];
            *reinterpret_cast<uint4*>(&Vs[key * LD + pc * E]) = vreg[n];
        }
    };

    const auto fmax2 = [](float a, float b) { return fmaxf(a, b); };
    const auto fadd2 = [](float a, float b) { return a + b; };
    const int kb0 = tl.k0 & ~(KB - 1);
    fetch(kb0);
    for (int kb = kb0; kb < tl.k1; kb += KB) {
        __syncthreads();  // every wave is done with the previous block's image
        stage();
        __syncthreads();
        if (kb + KB < tl.k1) fetch(kb + KB);
        if (!(kb < whi && kb + KB > wlo)) continue;  // wave-uniform: no query of this wave sees the block
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            // scores of the block's 2 x 16 keys for query r: s[kt][i] = key kb + 16 kt + 4 gq + i
            f32x4_t s[2];
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                s[kt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                const T* krow = &Ks[(kt * 16 + r) * LD];
                if constexpr (BF) {
#pragma unroll
                    for (int j = 0; j < NQ; ++j) {
                        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(krow + 32 * j + 8 * gq);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qh[h][j], s[kt], 0, 0, 0);
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[h][j], s[kt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
                        s[kt] = __builtin_amdgcn_mfma_f32_16x16x4f32(krow[4 * j + gq], qf[h][j], s[kt], 0, 0, 0);
                }
            }
            float sv[8], mx = -INFINITY;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pos = kb + kt * 16 + 4 * gq + i;
                    const bool ok = pos >= cs && pos < ce;
                    sv[kt * 4 + i] = ok ? s[kt][i] * scale : -INFINITY;
                    mx = fmaxf(mx, sv[kt * 4 + i]);
                }
            mx = quad_rows(mx, fmax2);
            const float mnew = fmaxf(mrun[h], mx);
            const bool none = mnew == -INFINITY;  // nothing seen so far: keep the empty state
            const float alpha = none ? 1.0f : expf(mrun[h] - mnew);
            float p[8], ps = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                p[e] = sv[e] == -INFINITY ? 0.f : expf(sv[e] - mnew);
                ps += p[e];
            }
            ps = quad_rows(ps, fadd2);
            lrun[h] = lrun[h] * alpha + ps;
            mrun[h] = mnew;
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) acc[h][dt] *= alpha;
            // O^T += V^T P^T
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                if constexpr (BF) {
                    s16x4_t ph, pl;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u16 hb = f32_to_bf16_bits(p[kt * 4 + e]);
                        ph[e] = (short)hb;
                        pl[e] = (short)f32_to_bf16_bits(p[kt * 4 + e] - bf16_bits_to_f32(hb));
                    }
                    // transposing read: lane 4 q' + p' of a 16-lane group points at key row 4 gq + q', columns 4 p' ..; it receives
                    // column (lane & 15) of the four rows
                    const T* vrow = &Vs[(kt * 16 + 4 * gq + (r >> 2)) * LD + 4 * (r & 3)];
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt) {
                        const s16x4_t vt = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) s16x4_t*)(vrow + dt * 16));
                        acc[h][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, ph, acc[h][dt], 0, 0, 0);
                        acc[h][dt] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vt, pl, acc[h][dt], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int dt = 0; dt < NDT; ++dt)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[h][dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(to_f32<T>(Vs[(kt * 16 + 4 * gq + j) * LD + dt * 16 + r]), p[kt * 4 + j],
                                                                              acc[h][dt], 0, 0, 0);
                }
            }
        }
    }

    if (!vq) return;
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        T* op = out + ((long)(tl.m0 + qi) * nh + (long)g * REP + h) * D + 4 * gq;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = lrun[h] > 0.f ? acc[h][dt][i] / lrun[h] : 0.f;
            store4_split<T>(op + dt * 16, out_lo, o);
        }
    }
}

void build_attn_tiles(const Tok* toks, int M, std::vector<AttnTile>& tiles) {
    tiles.clear();
    for (int m = 0; m < M;) {
        AttnTile t{m, 1, toks[m].row, toks[m].ctx_start, toks[m].ctx_start + toks[m].ctx_len};
        int e = m + 1;
        while (e < M && t.count < ATTN_TILE_TOKENS && toks[e].row == t.row && toks[e].pos == toks[e - 1].pos + 1) {
            t.k0 = std::min(t.k0, toks[e].ctx_start);
            t.k1 = std::max(t.k1, toks[e].ctx_start + toks[e].ctx_len);
            ++t.count;
            ++e;
        }
        tiles.push_back(t);
        m = e;
    }
}

bool attention_tiles_supported(int d, int nh, int nkv) {
    return (d == 64 || d == 128) && nkv >= 1 && nh % nkv == 0 && (nh / nkv == 1 || nh / nkv == 2);
}

template <typename T>
static void launch_tiles_t(hipStream_t st, const float* q, const Tok* toks, const AttnTile* tiles, int ntiles, const KVView& kv, int layer,
                           int nh, float scale, T* out, long out_lo) {
    const int rep = nh / kv.nkv;
    dim3 grid(ntiles, kv.nkv), block(256);
#define QTTS_TILE_CASE(DD, RR)                                                                                                     \
    if (kv.d == DD && rep == RR) {                                                                                                 \
        hipLaunchKernelGGL((attention_tile_kernel<T, DD, RR>), grid, block, 0, st, q, toks, tiles, kv, layer, nh, scale, out, out_lo); \
        return;                                                                                                                    \
    }
    QTTS_TILE_CASE(64, 1)
    QTTS_TILE_CASE(64, 2)
    QTTS_TILE_CASE(128, 1)
    QTTS_TILE_CASE(128, 2)
#undef QTTS_TILE_CASE
    throw Error(ERR_SHAPE, "tile attention: unsupported head shape");
}

void launch_attention_tiles(hipStream_t st, int dtype, const float* q, const Tok* toks, const AttnTile* tiles, int ntiles, const KVView& kv,
                            int layer, int nh, float scale, void* out, long out_lo) {
    QTTS_CHECK(attention_tiles_supported(kv.d, nh, kv.nkv), ERR_SHAPE, "tile attention: unsupported head shape");
    QTTS_CHECK(ntiles >= 1 && tiles && toks && q && out, ERR_BAD_ARG, "tile attention: missing operands");
    if (dtype == BF16) {
        launch_tiles_t<bf16_t>(st, q, toks, tiles, ntiles, kv, layer, nh, scale, (bf16_t*)out, out_lo);
    } else {
        QTTS_CHECK(out_lo == 0, ERR_BAD_ARG, "attention: a remainder plane exists for bf16 activations only");
        launch_tiles_t<float>(st, q, toks, tiles, ntiles, kv, layer, nh, scale, (float*)out, 0);
    }
}

}  // namespace qtts
