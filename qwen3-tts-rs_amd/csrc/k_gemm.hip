// Weight-streaming skinny GEMM for gfx950: out[ks][m][n] = sum_{k in slice ks} X[m][k] * W[n][k].
//
// Replaces every candle_nn::Linear on the decode path (qwen_tts/src/nn/attention/unified.rs:189-191,
// 242; nn/mlp.rs:67,72; nn/talker.rs:402; nn/code_predictor.rs:443,464,514).  The weight matrix is read
// exactly once from HBM; M (tokens in flight) is at most 64 per z-slice, so the kernel is HBM-bound
// and built around keeping many 16-byte loads in flight per lane rather than around LDS tiling.
//
// Work split: block = 4 waves; a block owns NT 16-row tiles of W for all M rows of one z-slice;
// the K range of split `ks` is cut into 4 contiguous wave slices; each wave runs
// v_mfma_f32_16x16x32_bf16 (or the exact v_mfma_f32_16x16x4_f32 for f32 weights) with the W tile as
// the A operand and the X tile as the B operand, so a lane ends up holding four consecutive n for
// one m.  The four wave partials are summed through LDS in wave order (deterministic).
#include "conv_epi.hpp"
#include <cstdlib>
#include <algorithm>

#ifndef QTTS_NT_WEIGHTS
#define QTTS_NT_WEIGHTS 1
#endif

namespace qtts {

// Tuning switches, read from the environment once per process (the first call from any thread fixes them for all).
struct GemmEnv {
    int force_kind, force_nt, force_nch, min_blocks;
    bool xcd_order, conv_halo_off;
    int mt_min;  // smallest row tile count per block the decode GEMM plans may choose (1, 2 or 4)
};
static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static const GemmEnv& gemm_env() {
    static const GemmEnv e{env_int("QTTS_GEMM_KIND", -1), env_int("QTTS_GEMM_NT", 0), env_int("QTTS_GEMM_NCH", 0),
                           env_int("QTTS_GEMM_MIN_BLOCKS", 192), env_int("QTTS_GEMM_XCD", 1) != 0, env_int("QTTS_CONV_HALO", 1) == 0,
                           env_int("QTTS_GEMM_MT_MIN", 4)};
    return e;
}

template <typename T>
struct Frag;
template <>
struct Frag<bf16_t> {
    static constexpr int CHUNK = 64;  // K elements covered by one wave-wide 32-byte-per-lane load
    static constexpr int EPL = 16;    // elements per lane
    bf16x8_t v[2];
    // hs: elements between the lane's two 16-byte pieces (8 in a row-major matrix, 64 * 8 in a packed weight tile)
    __device__ __forceinline__ void load(const bf16_t* p, int hs = 8) {
        v[0] = *reinterpret_cast<const bf16x8_t*>(p);
        v[1] = *reinterpret_cast<const bf16x8_t*>(p + hs);
    }
    // weights that one block reads once: non-temporal, so that they do not displace the activations in L2
    __device__ __forceinline__ void load_stream(const bf16_t* p, int hs = 8) {
        if (QTTS_NT_WEIGHTS) {
            v[0] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p));
            v[1] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p + hs));
        } else {
            load(p, hs);
        }
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], b.v[1], c, 0, 0, 0);
        return c;
    }
};
template <>
struct Frag<float> {
    static constexpr int CHUNK = 32;
    static constexpr int EPL = 8;
    float v[8];
    __device__ __forceinline__ void load(const float* p, int hs = 4) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + hs);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    __device__ __forceinline__ void load_stream(const float* p, int hs = 4) {
        if (QTTS_NT_WEIGHTS) {
            const f32x4_t a = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
            const f32x4_t b = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p + hs));
            v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
        } else {
            load(p, hs);
        }
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
        return c;
    }
};

// This lane's fragment of the 16-row weight tile that starts at row n0 (a multiple of 16): chunk c, piece h is at
// p + c * cs + h * hs elements, in a row-major matrix or a packed one (WMat).
template <typename T>
struct WTile {
    const T* p;
    int cs, hs;
    __device__ __forceinline__ WTile() : p(nullptr), cs(0), hs(0) {}
    __device__ __forceinline__ WTile(const T* W, int n0, int K, int lane, int pk) {
        constexpr int HALF = 16 / sizeof(T);
        if (pk) {
            cs = 128 * HALF;
            hs = 64 * HALF;
            p = W + (size_t)(n0 >> 4) * (K / Frag<T>::CHUNK) * cs + lane * HALF;
        } else {
            cs = Frag<T>::CHUNK;
            hs = HALF;
            p = W + (size_t)(n0 + (lane & 15)) * K + (lane >> 4) * Frag<T>::EPL;
        }
    }
    __device__ __forceinline__ const T* chunk(int c) const { return p + (size_t)c * cs; }
};

// Asynchronous copy of one 16-row x 128-byte activation unit (rows `row0`.., row stride ldx elements, starting at element
// column `col`) into the 2 KiB LDS unit `dst`, as two DMA instructions issued by one wave (device_utils.hpp: unit_*).
template <typename T>
__device__ __forceinline__ void dma_unit_half(const T* X, long row0, long ldx, long col, int h2, uint4* dst, int lane) {
    constexpr int HALF = 16 / sizeof(T);
    const T* src = X + (row0 + unit_src_row(lane, h2)) * ldx + col + unit_src_piece(lane, h2) * HALF;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(dst + h2 * 64), 16, 0, 0);
}
// the MFMA operand of lane (r, g) out of an LDS unit
template <typename T>
__device__ __forceinline__ Frag<T> unit_frag(const uint4* unit, int s0, int s1) {
    Frag<T> b;
    const uint4 h0 = unit[s0];
    const uint4 h1 = unit[s1];
    __builtin_memcpy(&b, &h0, 16);
    __builtin_memcpy((char*)&b + 16, &h1, 16);
    return b;
}

// LO: X carries a second plane `x_lo` elements behind it (the rounding remainder of the activations, see
// device_utils.hpp store_split); both planes are multiplied, so the product sees the activations to ~16 bits.
template <typename T, int MT, int NT, bool LO>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const T* __restrict__ W, const T* __restrict__ X,
                                                          float* __restrict__ out, int N, int K, int KS, int ldx,
                                                          long out_ks_stride, long x_lo, int wpk) {
    using F = Frag<T>;
    __shared__ f32x4_t red[4][MT * NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int ks = blockIdx.y;
    const int row0 = blockIdx.z * (16 * MT);
    const int nC = K / F::CHUNK, S = KS * 4, s = ks * 4 + wave;
    const int c0 = (int)((long)s * nC / S), c1 = (int)((long)(s + 1) * nC / S);

    WTile<T> wp[NT];
    const T* xp[MT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = WTile<T>(W, n0 + 16 * j, K, lane, wpk);
#pragma unroll
    for (int i = 0; i < MT; ++i) xp[i] = X + (size_t)(row0 + 16 * i + r) * ldx + g * F::EPL;

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};


    // two register stages: the loads of chunk c+1 are in flight while chunk c feeds the MFMAs
    constexpr int MB = LO ? 2 * MT : MT;  // X fragments per stage: the lo plane's follow the hi plane's
    F a0[NT], b0[MB], a1[NT], b1[MB];
    auto load_stage = [&](F(&a)[NT], F(&b)[MB], int c) {
        const size_t off = (size_t)c * F::CHUNK;
#pragma unroll
        for (int j = 0; j < NT; ++j) a[j].load_stream(wp[j].chunk(c), wp[j].hs);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[i].load(xp[i] + off);
        if constexpr (LO) {
#pragma unroll
            for (int i = 0; i < MT; ++i) b[MT + i].load(xp[i] + x_lo + off);
        }
    };
    auto mma_stage = [&](const F(&a)[NT], const F(&b)[MB]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                acc[i][j] = F::mma(a[j], b[i], acc[i][j]);
                if constexpr (LO) acc[i][j] = F::mma(a[j], b[MT + i], acc[i][j]);
            }
    };
    if (c0 < c1) load_stage(a0, b0, c0);
    for (int c = c0; c < c1; c += 2) {
        if (c + 1 < c1) load_stage(a1, b1, c + 1);
        mma_stage(a0, b0);
        if (c + 2 < c1) load_stage(a0, b0, c + 2);
        if (c + 1 < c1) mma_stage(a1, b1);
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) red[wave][i * NT + j][lane] = acc[i][j];
    __syncthreads();
    float* o = out + (size_t)ks * out_ks_stride;
    for (int idx = threadIdx.x; idx < MT * NT * 64; idx += 256) {
        const int tile = idx >> 6, l = idx & 63;
        f32x4_t v = red[0][tile][l];
        v += red[1][tile][l];
        v += red[2][tile][l];
        v += red[3][tile][l];
        const int i = tile / NT, j = tile % NT;
        const int m = row0 + 16 * i + (l & 15);
        const int n = n0 + 16 * j + 4 * (l >> 4);
        *reinterpret_cast<f32x4_t*>(o + (size_t)m * N + n) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// Variant for M >= 32: the four waves of a block own different 16*NT-row slices of W and SHARE the X
// tile through LDS, so X is read from L2 once per 64*NT weight rows instead of once per 16*NT.
// Single phase, no loop-carried pipeline: every W load of the block's K range (NCH chunks) is issued up
// front straight to registers, X fragment blocks are DMA'd into LDS (lane-linear 1 KiB blocks laid
// out exactly as the MFMA B operand wants them, so the later ds_read_b128 are conflict-free), one
// barrier, then the MFMAs run out of registers + LDS.  K is split across blocks only (grid.y).
template <typename T, int MT, int NT, int NCH, bool LO>
__global__ __launch_bounds__(256) void gemm_ldsx_kernel(const T* __restrict__ W, const T* __restrict__ X,
                                                        float* __restrict__ out, int N, int K, int ldx,
                                                        long out_ks_stride, int GX, int KS, int Z, long x_lo, int wpk) {
    using F = Frag<T>;
    constexpr int NFB = NCH * MT * 2;      // 1 KiB half units of one plane
    __shared__ uint4 xs[(LO ? 2 : 1) * NFB * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (Z > 1) {
        // Several row slices (1-D launch): workgroups go to the 8 XCDs round-robin by linear id, so the Z slices of one weight
        // tile are given ids 8 apart inside one run of 8 * Z ids - same XCD, same moment - and share the tile through that L2.
        const int L = blockIdx.x;
        const int t = (L / (8 * Z)) * 8 + (L & 7);
        if (t >= GX * KS) return;
        bz = (L >> 3) % Z;
        bx = t % GX;
        by = t / GX;
    }
    const int n0 = bx * (64 * NT) + wave * (16 * NT);
    const int ks = by;
    const int row0 = bz * (16 * MT);
    const size_t kbeg = (size_t)ks * NCH * F::CHUNK;

    // X units -> LDS (asynchronous, no VGPRs); a remainder plane's units follow the first plane's in LDS
    auto dma_plane = [&](int pl) {
        for (int f = wave; f < NFB; f += 4) {
            const int c = f / (MT * 2), i = (f >> 1) % MT, h2 = f & 1;
            dma_unit_half<T>(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)kbeg + (long)c * F::CHUNK, h2, xs + (pl * NFB + (f & ~1)) * 64, lane);
        }
    };
    dma_plane(0);
    // all W loads of this block's K range
    F a[NCH][NT];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            // one row slice: every weight byte is read by exactly one block (streamed); several: the slices share it through L2
            const WTile<T> wt(W, n0 + 16 * j, K, lane, wpk);
            const T* wsrc = wt.chunk(ks * NCH + c);
            if (Z > 1) a[c][j].load(wsrc, wt.hs);
            else a[c][j].load_stream(wsrc, wt.hs);
        }
    if constexpr (LO) dma_plane(1);

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's LDS DMA has landed before it signals the barrier
    __syncthreads();
    const int s0 = unit_slot(r, g, 0), s1 = unit_slot(r, g, 1);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int pl = 0; pl < (LO ? 2 : 1); ++pl) {
                const F b = unit_frag<T>(xs + (pl * NFB + (c * MT + i) * 2) * 64, s0, s1);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[c][j], b, acc[i][j]);
            }
        }
    }
    float* o = out + (size_t)ks * out_ks_stride;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4_t*>(o + (size_t)(row0 + 16 * i + r) * N + n0 + 16 * j + 4 * g) = acc[i][j];
}

template <typename T, int MT, int NT, int NCH, bool LO>
static void launch_ldsx(hipStream_t st, const T* W, int wpk, const T* X, float* out, int Mpad, int N, int K, int ldx, long x_lo) {
    const int KS = K / (NCH * Frag<T>::CHUNK);
    const int GX = N / (64 * NT), Z = Mpad / (16 * MT);
    const bool xcd_order = gemm_env().xcd_order;
    if (Z > 1 && xcd_order) {
        const int groups = (GX * KS + 7) / 8;
        hipLaunchKernelGGL((gemm_ldsx_kernel<T, MT, NT, NCH, LO>), dim3(groups * 8 * Z), dim3(256), 0, st, W, X, out, N, K, ldx, (long)Mpad * N,
                           GX, KS, Z, x_lo, wpk);
        return;
    }
    dim3 grid(GX, KS, Z);
    hipLaunchKernelGGL((gemm_ldsx_kernel<T, MT, NT, NCH, LO>), grid, dim3(256), 0, st, W, X, out, N, K, ldx, (long)Mpad * N, GX, KS, 1, x_lo, wpk);
}

template <typename T, int MT, bool LO>
static bool launch_ldsx_mt(hipStream_t st, const T* W, int wpk, const T* X, float* out, int Mpad, int N, int K, int ldx, int NT, int NCH, long x_lo) {
#define QTTS_LDSX(nt, nch) if (NT == nt && NCH == nch) { launch_ldsx<T, MT, nt, nch, LO>(st, W, wpk, X, out, Mpad, N, K, ldx, x_lo); return true; }
    QTTS_LDSX(1, 4) QTTS_LDSX(1, 8) QTTS_LDSX(2, 4) QTTS_LDSX(2, 8)
    if constexpr (MT <= 2) { QTTS_LDSX(1, 16) }
#undef QTTS_LDSX
    return false;
}

template <typename T, int MT, bool LO>
static void launch_mt(hipStream_t st, const T* W, int wpk, const T* X, float* out, int Mpad, int N, int K, int KS, int ldx, long x_lo) {
    const int Z = Mpad / (16 * MT);
    const long kstride = (long)Mpad * N;
    if (N % 32 == 0 && (long)(N / 32) * KS * Z >= 256) {
        dim3 grid(N / 32, KS, Z);
        hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 2, LO>), grid, dim3(256), 0, st, W, X, out, N, K, KS, ldx, kstride, x_lo, wpk);
    } else {
        dim3 grid(N / 16, KS, Z);
        hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 1, LO>), grid, dim3(256), 0, st, W, X, out, N, K, KS, ldx, kstride, x_lo, wpk);
    }
}

template <typename T, bool LO>
static void launch_t(hipStream_t st, const T* W, int wpk, const T* X, float* out, int Mpad, int N, int K, int KS, int ldx, long x_lo) {
    if (Mpad == 16) launch_mt<T, 1, LO>(st, W, wpk, X, out, Mpad, N, K, KS, ldx, x_lo);
    else if (Mpad == 32) launch_mt<T, 2, LO>(st, W, wpk, X, out, Mpad, N, K, KS, ldx, x_lo);
    else launch_mt<T, 4, LO>(st, W, wpk, X, out, Mpad, N, K, KS, ldx, x_lo);
}

// ------------------------------------------------------------------------------------------------
// Large-M variant (codec / speaker convolutions and transformer): rows = time positions, weights are
// small and L2-resident, K is walked in phases of NCH chunks inside the block (no split-K slabs).
// The X operand supports convolution addressing without im2col: with channels-last activations the
// taps of a causal conv are row offsets, so chunk cg reads X[(m + (cg / cpt) * tap_rows)][(cg % cpt) * CHUNK ...]
// (cpt = chunks per tap).  A plain linear layer is the single-tap case.
// With EPI the conv epilogue (conv_epi.hpp) runs on the accumulators in registers and no f32 output is written.
template <typename T, int NT, int NCH, bool EPI, bool LO>
__global__ __launch_bounds__(256) void gemm_rows_kernel(const T* __restrict__ W, const T* __restrict__ X,
                                                        float* __restrict__ out, int N, int K, int ldx, int cpt,
                                                        long tap_stride, long row_base, ConvEpi e, long x_lo, int wpk) {
    using F = Frag<T>;
    constexpr int MT = 4;
    constexpr int NFB = NCH * MT * 2;
    __shared__ uint4 xs[(LO ? 2 : 1) * NFB * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (64 * NT) + wave * (16 * NT);
    const long row0 = row_base + (long)blockIdx.y * 64;
    const int nC = K / F::CHUNK;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int s0 = unit_slot(r, g, 0), s1 = unit_slot(r, g, 1);
    WTile<T> wt[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wt[j] = WTile<T>(W, n0 + 16 * j, K, lane, wpk);
    for (int c0 = 0; c0 < nC; c0 += NCH) {
        for (int fb = wave; fb < (LO ? 2 : 1) * NFB; fb += 4) {
            const int pl = fb / NFB, f = fb % NFB;
            const int c = f / (MT * 2), i = (f >> 1) % MT, h2 = f & 1;
            const int cg = c0 + c;
            dma_unit_half<T>(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)(cg / cpt) * tap_stride + (long)(cg % cpt) * F::CHUNK, h2,
                             xs + (fb & ~1) * 64, lane);
        }
        F a[NCH][NT];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < NT; ++j) a[c][j].load(wt[j].chunk(c0 + c), wt[j].hs);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's LDS DMA has landed before it signals the barrier
        __syncthreads();
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const F b = unit_frag<T>(xs + ((c * MT + i) * 2) * 64, s0, s1);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[c][j], b, acc[i][j]);
                if constexpr (LO) {
                    const F bl = unit_frag<T>(xs + (NFB + (c * MT + i) * 2) * 64, s0, s1);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[c][j], bl, acc[i][j]);
                }
            }
        }
        __syncthreads();
    }
    if (EPI) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const long m = row0 + 16 * i + r;
            if (m >= e.M) continue;
            const long u = m / e.S_in;
            const int tp = (int)(m - u * e.S_in);
            if (tp >= e.lens[u]) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                conv_epi_apply<T>(e, u, tp, n0 + 16 * j + 4 * g, v);
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
            *reinterpret_cast<f32x4_t*>(out + (size_t)(row0 - row_base + 16 * i + r) * N + n0 + 16 * j + 4 * g) = acc[i][j];
}

template <typename T, bool EPI, bool LO = false>
static void launch_rows_t(hipStream_t st, const T* W, int wpk, const T* X, float* out, long Mpad, int N, int K, int ldx, int cpt, long tap_stride,
                          const ConvEpi& e, long x_lo = 0) {
    const int nC = K / Frag<T>::CHUNK;
    // out rows are relative to the launch: split launches at the grid.y limit
    for (long base = 0; base < Mpad; base += 65535L * 64) {
        const long rows = std::min<long>(Mpad - base, 65535L * 64);
        float* o = EPI ? nullptr : out + (size_t)base * N;
        const int gy = (int)(rows / 64);
#define QTTS_ROWS(nt, nch)                                                                                                          \
    hipLaunchKernelGGL((gemm_rows_kernel<T, nt, nch, EPI, LO>), dim3(N / (64 * nt), gy), dim3(256), 0, st, W, X, o, N, K, ldx, cpt, \
                       tap_stride, base, e, x_lo, wpk)
        // QTTS_ROWS_WIDE=0: two column tiles per wave for every shape
        static const bool wide = !(getenv("QTTS_ROWS_WIDE") && atoi(getenv("QTTS_ROWS_WIDE")) == 0);
        if (wide && !EPI && LO && N % 256 == 0 && nC % 2 == 0) {
            // two activation planes: with two column tiles per wave the LDS reads of the X fragments take as long as the MFMAs they
            // feed; four column tiles per wave halve them (64 x 256 block tiles; the 1.7B talker's prefill: 32.3 -> 27.5 ms).  The
            // choice depends on the layer, never on the row count.
            QTTS_ROWS(4, 2);
        } else if (N % 128 == 0) {
            if (nC % 4 == 0) QTTS_ROWS(2, 4);
            else if (nC % 3 == 0) QTTS_ROWS(2, 3);
            else if (nC % 2 == 0) QTTS_ROWS(2, 2);
            else QTTS_ROWS(2, 1);
        } else {
            if (nC % 4 == 0) QTTS_ROWS(1, 4);
            else if (nC % 3 == 0) QTTS_ROWS(1, 3);
            else if (nC % 2 == 0) QTTS_ROWS(1, 2);
            else QTTS_ROWS(1, 1);
        }
#undef QTTS_ROWS
    }
}

// out[m][n] = sum over taps/channels; M rows padded up to 64 must be readable in X (including the tap offsets).
void launch_gemm_rows(hipStream_t st, int dtype, WMat Wm, const void* X, float* out, long M, int N, int K, int ldx, int taps,
                      long tap_stride, long x_lo) {
    const void* W = Wm.p;
    const int wpk = Wm.packed ? 1 : 0;
    QTTS_CHECK(M >= 1 && N % 64 == 0 && taps >= 1 && K % taps == 0, ERR_SHAPE, "gemm_rows: N must be a multiple of 64");
    const int chunk = dtype == BF16 ? 64 : 32;
    QTTS_CHECK((K / taps) % chunk == 0, ERR_SHAPE, "gemm_rows: channels per tap must be a multiple of the K chunk");
    const long Mpad = (M + 63) / 64 * 64;
    const int cpt = K / taps / chunk;
    const ConvEpi none{};
    QTTS_CHECK(x_lo == 0 || dtype == BF16, ERR_BAD_ARG, "gemm_rows: a remainder plane exists for bf16 activations only");
    if (dtype == BF16 && x_lo) launch_rows_t<bf16_t, false, true>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, out, Mpad, N, K, ldx, cpt, tap_stride, none, x_lo);
    else if (dtype == BF16) launch_rows_t<bf16_t, false>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, out, Mpad, N, K, ldx, cpt, tap_stride, none);
    else launch_rows_t<float, false>(st, (const float*)W, wpk, (const float*)X, out, Mpad, N, K, ldx, cpt, tap_stride, none);
}

// Same GEMM with the conv epilogue applied in registers (no f32 output): e.acc is ignored, e.M bounds the rows.
void launch_gemm_rows_epi(hipStream_t st, int dtype, WMat Wm, const void* X, int K, int ldx, int taps, long tap_stride,
                          const ConvEpi& e, long x_lo) {
    const void* W = Wm.p;
    const int wpk = Wm.packed ? 1 : 0;
    const long M = e.M;
    const int N = e.N;
    QTTS_CHECK(M >= 1 && N % 64 == 0 && taps >= 1 && K % taps == 0, ERR_SHAPE, "gemm_rows: N must be a multiple of 64");
    QTTS_CHECK(e.N % 4 == 0 && e.Cout % 4 == 0 && e.N == e.r * e.Cout && e.Cvalid % 4 == 0 && e.lens && e.S_in >= 1 && conv_epi_vector_ok(e), ERR_SHAPE,
               "gemm_rows: bad epilogue shape");
    QTTS_CHECK(e.out2 == nullptr || e.act2 == 1 || (e.alpha_exp && e.beta_den), ERR_BAD_ARG, "gemm_rows: snake output without parameters");
    const int chunk = dtype == BF16 ? 64 : 32;
    QTTS_CHECK((K / taps) % chunk == 0, ERR_SHAPE, "gemm_rows: channels per tap must be a multiple of the K chunk");
    QTTS_CHECK(x_lo == 0 || dtype == BF16, ERR_BAD_ARG, "gemm_rows: a remainder plane exists for bf16 activations only");
    if (!gemm_env().conv_halo_off && launch_conv_halo(st, dtype, Wm, X, K, ldx, taps, tap_stride, e, x_lo)) return;
    const long Mpad = (M + 63) / 64 * 64;
    const int cpt = K / taps / chunk;
    if (dtype == BF16 && x_lo) launch_rows_t<bf16_t, true, true>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, nullptr, Mpad, N, K, ldx, cpt, tap_stride, e, x_lo);
    else if (dtype == BF16) launch_rows_t<bf16_t, true>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, nullptr, Mpad, N, K, ldx, cpt, tap_stride, e);
    else launch_rows_t<float, true>(st, (const float*)W, wpk, (const float*)X, nullptr, Mpad, N, K, ldx, cpt, tap_stride, e);
}

int gemm_pad_rows(int M) { return M <= 16 ? 16 : (M <= 32 ? 32 : (M + 63) / 64 * 64); }

// Picks the kernel variant and the split-K factor for a shape.  Deterministic in (M, N, K, dtype) so that the
// consumers of the partial slabs can recompute KS.
//
// LDS-shared-X kernel: a block owns 64*NT weight rows x NCH K-chunks for a slice of 16*MT token rows.  Slab traffic
// (KS * M * N * 4 bytes written, then read back by the consumer) falls with NCH; the block count (N / (64 NT)) * KS * Z,
// Z = Mpad / (16 MT), is kept up by cutting the rows into more slices instead of the K range into more slabs: the slices of
// one weight tile run on one XCD and share the tile through its L2, so HBM still streams every weight byte once.
GemmPlan gemm_plan(int M, int N, int K, int dtype) {
    const GemmEnv& env = gemm_env();
    const int force_kind = env.force_kind, force_nt = env.force_nt, force_nch = env.force_nch;
    // first candidate (fewest slabs first) that still gives this many blocks
    const int min_blocks = env.min_blocks;
    const int chunk = dtype == BF16 ? 64 : 32;
    const int nC = K / chunk;
    const int Mpad = gemm_pad_rows(M);
    GemmPlan p{1, 0, 1, 0, 4};
    if (Mpad >= 32 && force_kind != 0) {
        const int mt_hi = Mpad == 32 ? 2 : 4;
        const int mt_lo = Mpad > 64 ? 4 : std::max(1, std::min(env.mt_min, mt_hi));
        long best_blocks = -1;
        for (int nch : {16, 8, 4}) {
            if (force_nch && nch != force_nch) continue;
            if (nC % nch != 0) continue;
            const int ks = nC / nch;
            if (ks > GEMM_MAX_KS) continue;
            for (int mt = mt_hi; mt >= mt_lo; mt >>= 1) {
                if (nch * mt > 32) continue;  // X tile (two planes in bf16 runs) within 128 KB of LDS
                for (int nt : {2, 1}) {
                    if (force_nt && nt != force_nt) continue;
                    if (nch * nt > 16) continue;  // weight fragments of the whole K range live in registers
                    if (N % (64 * nt) != 0) continue;
                    const long blocks = (long)(N / (64 * nt)) * ks * (Mpad / (16 * mt));
                    if (blocks >= min_blocks) return GemmPlan{ks, 1, nt, nch, mt};
                    if (blocks > best_blocks) { best_blocks = blocks; p = GemmPlan{ks, 1, nt, nch, mt}; }
                }
            }
        }
        if (best_blocks >= 128 || force_kind == 1) { if (p.kind == 1) return p; }
        p = GemmPlan{1, 0, 1, 0, 4};
    }
    const int Z = Mpad <= 32 ? 1 : Mpad / 64;
    const long nblk = (long)(N / 16) * Z;  // blocks at KS = 1 with one 16-row tile each
    int ks = 1;
    // aim for >= ~512 blocks (2 per CU) while every wave keeps at least two chunks
    while (nblk * ks < 512 && nC / (4 * ks * 2) >= 2 && ks < GEMM_MAX_KS) ks *= 2;
    p.KS = ks;
    p.kind = 0;
    p.NT = (N % 32 == 0 && (long)(N / 32) * ks * Z >= 256) ? 2 : 1;
    return p;
}

int gemm_pick_ks(int M, int N, int K, int dtype) { return gemm_plan(M, N, K, dtype).KS; }

template <typename T, bool LO>
static void launch_plan(hipStream_t st, const GemmPlan& p, const T* W, int wpk, const T* X, float* out, int Mpad, int N, int K, int ldx, long x_lo) {
    if (p.kind == 1) {
        bool ok = false;
        if (p.MT == 1) ok = launch_ldsx_mt<T, 1, LO>(st, W, wpk, X, out, Mpad, N, K, ldx, p.NT, p.NCH, x_lo);
        else if (p.MT == 2) ok = launch_ldsx_mt<T, 2, LO>(st, W, wpk, X, out, Mpad, N, K, ldx, p.NT, p.NCH, x_lo);
        else ok = launch_ldsx_mt<T, 4, LO>(st, W, wpk, X, out, Mpad, N, K, ldx, p.NT, p.NCH, x_lo);
        QTTS_CHECK(ok, ERR_STATE, "gemm: no kernel instance for the planned variant");
        return;
    }
    launch_t<T, LO>(st, W, wpk, X, out, Mpad, N, K, p.KS, ldx, x_lo);
}

void launch_gemm(hipStream_t st, int dtype, WMat Wm, const void* X, float* out, int M, int N, int K, int KS, int ldx, long x_lo) {
    const void* W = Wm.p;
    const int wpk = Wm.packed ? 1 : 0;
    QTTS_CHECK(N % 16 == 0, ERR_SHAPE, "gemm: N must be a multiple of 16");
    QTTS_CHECK(K % 64 == 0, ERR_SHAPE, "gemm: K must be a multiple of 64");
    QTTS_CHECK(KS >= 1 && M >= 1, ERR_SHAPE, "gemm: bad M/KS");
    QTTS_CHECK(x_lo == 0 || dtype == BF16, ERR_BAD_ARG, "gemm: a remainder plane exists for bf16 activations only");
    const int Mpad = gemm_pad_rows(M);
    GemmPlan p = gemm_plan(M, N, K, dtype);
    if (p.KS != KS) p = GemmPlan{KS, 0, 1, 0, 4};  // caller forced a split factor: generic kernel
    if (dtype == BF16 && x_lo) launch_plan<bf16_t, true>(st, p, (const bf16_t*)W, wpk, (const bf16_t*)X, out, Mpad, N, K, ldx, x_lo);
    else if (dtype == BF16) launch_plan<bf16_t, false>(st, p, (const bf16_t*)W, wpk, (const bf16_t*)X, out, Mpad, N, K, ldx, 0);
    else launch_plan<float, false>(st, p, (const float*)W, wpk, (const float*)X, out, Mpad, N, K, ldx, 0);
}

// ------------------------------------------------------------------------------------------------
// Row-major weight matrix -> fragment order (WMat), 16 bytes per thread.
__global__ __launch_bounds__(256) void pack_weight_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long pieces, int ppr) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;   // source piece: row n, piece q of the row
    if (i >= pieces) return;
    const long n = i / ppr;
    const int q = (int)(i - n * ppr);
    const int c = q >> 3, pp = q & 7;
    const long nC = ppr >> 3;
    const long d = ((((n >> 4) * nC + c) * 2 + (pp & 1)) * 64) + (n & 15) + 16 * (pp >> 1);
    dst[d] = src[i];
}
void launch_pack_weight(hipStream_t st, const void* src, void* dst, int N, long row_bytes) {
    QTTS_CHECK(N >= 16 && N % 16 == 0 && row_bytes >= 128 && row_bytes % 128 == 0, ERR_SHAPE, "pack_weight: N must be a multiple of 16 and rows of 128 bytes");
    QTTS_CHECK(src != dst, ERR_BAD_ARG, "pack_weight: in-place packing is not supported");
    const long pieces = (long)N * (row_bytes / 16);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, st, (const uint4*)src, (uint4*)dst, pieces,
                       (int)(row_bytes / 16));
}

}  // namespace qtts
