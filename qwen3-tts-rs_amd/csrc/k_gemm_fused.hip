// Decode-path GEMMs that own finished outputs and apply their consumer's element-wise work themselves (bf16 stacks).
//   full_k : hidden size K <= 2048.  A block owns 64 weight rows (four waves x 16, or eight waves x half the K range, all their
//            K in registers) for 16 MT token rows, whose X tile (both planes of split activations) arrives once in LDS by DMA; the
//            rows are cut into slices so that the launch has about 200 blocks or more, and the slices of a weight tile share it
//            through L2.  Epilogues: PLAIN (bias, f32 out) for q|k|v and the heads, SWIGLU for gate|up.
//   k_split: long K, few outputs (N = hidden).  A block owns 16 weight rows x 16 token rows; its eight waves take contiguous
//            eighths of K - each DMAs its own part of the X tile to LDS and reads its weights straight to registers - and sum
//            their partial tiles through LDS in wave order.  Epilogue RESID: residual add + the next RMSNorm's weight.
// With both, a decoder layer whose hidden size fits full_k at 64 rows (the code predictor) is five launches instead of eight:
//   q|k|v (full_k, PLAIN)  ->  attention  ->  o_proj (k_split, RESID)  ->  gate|up (full_k, SWIGLU)  ->  down (k_split, RESID)
// (qwen_tts/src/nn/decoder_layer.rs:63-86, nn/attention/unified.rs:189-242, nn/mlp.rs:59-82) and no split-K slab is
// written or read back.  The RMSNorm in front of a projection (nn/norm.rs:17-23) is then split in two: the producer of the
// residual stream stores xb = T(x * norm_w) and per-tile partial sums of squares of x, the consuming GEMM multiplies its finished
// sums by 1 / sqrt(sum x^2 / H + eps) - the same value up to the order of two roundings.
// Weights are read in fragment order (WMat packed) or row-major; X is row-major.
#include "device_utils.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <string>

namespace qtts {

namespace {

struct BFrag {  // one wave-wide fragment of 64 K elements: 16 elements per lane as two 16-byte halves
    bf16x8_t v[2];
    __device__ __forceinline__ void load(const bf16_t* p, int hs) {
        v[0] = *reinterpret_cast<const bf16x8_t*>(p);
        v[1] = *reinterpret_cast<const bf16x8_t*>(p + hs);
    }
    __device__ __forceinline__ void load_stream(const bf16_t* p, int hs) {  // read-once weights: keep them out of the L2 the activations live in
        v[0] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p));
        v[1] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8_t*>(p + hs));
    }
};
__device__ __forceinline__ f32x4_t mma(const BFrag& a, const BFrag& b, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], b.v[1], c, 0, 0, 0);
    return c;
}

constexpr int CHUNK = 64, EPL = 16, HALF = 8;

// this lane's fragment of the 16-row weight tile at row n0: chunk c, piece h at p + c * cs + h * hs (row-major or packed matrix)
struct WTileB {
    const bf16_t* p;
    int cs, hs;
    __device__ __forceinline__ WTileB(const bf16_t* W, int n0, int K, int lane, int pk) {
        if (pk) {
            cs = 128 * HALF;
            hs = 64 * HALF;
            p = W + (size_t)(n0 >> 4) * (K / CHUNK) * cs + lane * HALF;
        } else {
            cs = CHUNK;
            hs = HALF;
            p = W + (size_t)(n0 + (lane & 15)) * K + (lane >> 4) * EPL;
        }
    }
};
// one half (rows 8 h2 ..) of a 16-row x 64-element unit of X -> LDS, row-contiguous and swizzled (device_utils.hpp unit_*)
__device__ __forceinline__ void dma_half(const bf16_t* X, long row0, long ldx, long col, int h2, uint4* unit, int lane) {
    const bf16_t* src = X + (row0 + unit_src_row(lane, h2)) * ldx + col + unit_src_piece(lane, h2) * HALF;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(unit + h2 * 64), 16, 0, 0);
}
__device__ __forceinline__ BFrag unit_frag(const uint4* unit, int s0, int s1) {
    BFrag b;
    const uint4 h0 = unit[s0];
    const uint4 h1 = unit[s1];
    __builtin_memcpy(&b, &h0, 16);
    __builtin_memcpy((char*)&b + 16, &h1, 16);
    return b;
}

// Sum of squares of every row of the block from the producer's per-tile partials, spread over the block's four waves so that
// the loads are in flight together (one thread summing 64 partials in a row would pay 64 L2 round trips): ssq[q][row] receives
// the partial sum of the tiles t = q mod 4; row_scale() folds the four in a fixed order.
__device__ __forceinline__ void row_sumsq(const FusedEpi& e, int row0, int rows, float* ssq) {
    const int q = (threadIdx.x >> 6) & 3, row = threadIdx.x & 63;  // four consecutive waves take part
    if (row < rows) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (e.ss_in) {
            const float* p = e.ss_in + row0 + row;
            int t = q;
            for (; t + 12 < e.ntiles_in; t += 16) {
                const float v0 = p[(long)t * e.ss_ld], v1 = p[(long)(t + 4) * e.ss_ld], v2 = p[(long)(t + 8) * e.ss_ld], v3 = p[(long)(t + 12) * e.ss_ld];
                a0 += v0; a1 += v1; a2 += v2; a3 += v3;
            }
            for (; t < e.ntiles_in; t += 4) a0 += p[(long)t * e.ss_ld];
        }
        ssq[q * 64 + row] = (a0 + a1) + (a2 + a3);
    }
}
__device__ __forceinline__ float row_scale(const FusedEpi& e, const float* ssq, int mi) {
    if (!e.ss_in) return 1.0f;
    const float ss = (ssq[mi] + ssq[64 + mi]) + (ssq[128 + mi] + ssq[192 + mi]);
    return 1.0f / sqrtf(ss / (float)e.H_in + e.eps);
}

// Epilogue of a finished 16 MT x 16 output tile: lane (r, g) holds, for row 16 i + r, the four outputs n0 + 4 g .. + 3.
template <int MT, int MODE>
__device__ __forceinline__ void finished_tile_epilogue(const f32x4_t (&acc)[MT], const FusedEpi& e, const float* ssq, int row0, int n0, int N, int r, int g) {
    const int n = n0 + 4 * g;
    if (MODE == FUSED_PLAIN) {
        float bias[4] = {0.f, 0.f, 0.f, 0.f};
        if (e.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(e.bias + n);
            bias[0] = b4.x; bias[1] = b4.y; bias[2] = b4.z; bias[3] = b4.w;
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mi = 16 * i + r;
            const float rs = row_scale(e, ssq, mi);
            const f32x4_t v = acc[i];
            *reinterpret_cast<float4*>(e.out_f32 + (size_t)(row0 + mi) * e.ld_out + n) =
                make_float4(v[0] * rs + bias[0], v[1] * rs + bias[1], v[2] * rs + bias[2], v[3] * rs + bias[3]);
        }
    } else {
        // SWIGLU: the weight rows of a 16-row tile are [4 gate, 4 up, 4 gate, 4 up] of eight consecutive channels, so the
        // lanes g = 0 / 2 hold gate values and their neighbours 16 lanes up the matching up values
        bf16_t* act = (bf16_t*)e.act;
        const int I = N / 2;
        const int ch = (n0 >> 1) + 4 * (g >> 1);  // first of this lane pair's four channels
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int mi = 16 * i + r;
            const float rs = row_scale(e, ssq, mi);
            float gv[4], uv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                gv[q] = acc[i][q] * rs;
                uv[q] = __shfl_xor(gv[q], 16, 64);
            }
            if ((g & 1) == 0) {
                // four consecutive channels: one 8-byte store per plane
                bf16_t* o = act + (size_t)(row0 + mi) * I + ch;
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = gv[q] / (1.0f + expf(-gv[q])) * uv[q];
                u16 hb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) hb[q] = f32_to_bf16_bits(y[q]);
                *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16));
                if (e.act_lo) {
                    u16 lb[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) lb[q] = f32_to_bf16_bits(y[q] - bf16_bits_to_f32(hb[q]));
                    *reinterpret_cast<uint2*>(o + e.act_lo) = make_uint2((unsigned)lb[0] | ((unsigned)lb[1] << 16), (unsigned)lb[2] | ((unsigned)lb[3] << 16));
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ full_k
// KH: the K range is cut into KH halves taken by separate waves (4 KH waves per block; 16 chunks of weights per wave at most)
// LO: X carries a remainder plane x_lo elements behind it; both planes are multiplied
template <int MT, int MODE, int KH, bool LO>
__global__ __launch_bounds__(256 * KH) void gemm_full_k_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, int N, int K, int ldx,
                                                               long x_lo, int wpk, FusedEpi e) {
    constexpr int MAXC = 16, PL = LO ? 2 : 1;
    QTTS_TS(0);
    extern __shared__ uint4 xs[];  // [PL][nC][MT] units of 2 KiB, then 4 x 64 partial sums of squares, then (KH == 2) 4 x MT partial tiles
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int ct = wave & 3, kh = wave >> 2;
    const int nC = K / CHUNK, nCw = nC / KH;
    const int n0 = blockIdx.x * 64 + ct * 16;
    const int row0 = blockIdx.y * (16 * MT);
    float* ssq = reinterpret_cast<float*>(xs + (size_t)PL * nC * MT * 128);
    f32x4_t* red = reinterpret_cast<f32x4_t*>(ssq + 256);

    const int nhu = PL * nC * MT * 2;
    for (int f = wave; f < nhu; f += 4 * KH) {
        const int u = f >> 1;                     // unit: plane-major, then chunk, then row tile
        const int pl = u / (nC * MT), c = (u / MT) % nC, i = u % MT;
        dma_half(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)c * CHUNK, f & 1, xs + (size_t)u * 128, lane);
    }
    BFrag a[MAXC];
    const WTileB wt(W, n0, K, lane, wpk);
    const bool shared_w = gridDim.y > 1;  // several row slices read this tile: leave it in L2 for them
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
        if (c < nCw) {
            const bf16_t* src = wt.p + (size_t)(kh * nCw + c) * wt.cs;
            if (shared_w) a[c].load(src, wt.hs);
            else a[c].load_stream(src, wt.hs);
        }
    if (threadIdx.x < 256) row_sumsq(e, row0, 16 * MT, ssq);

    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    QTTS_TS(1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's LDS DMA has landed before it signals the barrier
    QTTS_TS(2);
    __syncthreads();
    QTTS_TS(3);
    const int s0 = unit_slot(r, g, 0), s1 = unit_slot(r, g, 1);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nCw) {
            const int cg = kh * nCw + c;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                acc[i] = mma(a[c], unit_frag(xs + (size_t)(cg * MT + i) * 128, s0, s1), acc[i]);
                if constexpr (LO) acc[i] = mma(a[c], unit_frag(xs + (size_t)((nC + cg) * MT + i) * 128, s0, s1), acc[i]);
            }
        }
    }
    QTTS_TS(4);
    if constexpr (KH > 1) {  // the upper K slices hand their tiles to the lowest one, which adds them in slice order
        if (kh > 0) {
#pragma unroll
            for (int i = 0; i < MT; ++i) red[(((kh - 1) * 4 + ct) * MT + i) * 64 + lane] = acc[i];
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int q = 0; q < KH - 1; ++q)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[i] += red[((q * 4 + ct) * MT + i) * 64 + lane];
    }
    QTTS_TS(5);
    finished_tile_epilogue<MT, MODE>(acc, e, ssq, row0, n0, N, r, g);
    QTTS_TS(6);
}

// ------------------------------------------------------------------------------------------------ stream
// Wide GEMMs at 64 token rows (q|k|v, gate|up of a 2048-wide stack): the X tile of the whole K range does not fit LDS, so a block
// (four waves x 16 weight rows, 64 token rows) walks K in phases of PC chunks through a ring of four 32 KB LDS buffers: while
// phase p is multiplied, the DMAs of phases p + 1 .. p + 3 are under way and the weight fragments of later phases travel to
// registers.  (With two 64 KB buffers and the DMA one phase ahead every phase waited for its tile: 20.2 us against 17.9.)
// The compiler serialises an LDS read behind every outstanding LDS DMA (it cannot tell the two buffers apart), so the fragment
// reads are issued by hand (ds_read_b128 in inline assembly, waited for with counted s_waitcnt lgkmcnt) and the barriers are
// bare s_barrier instructions behind counted s_waitcnt vmcnt.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifndef QTTS_STREAM_DBG
#define QTTS_STREAM_DBG 0  // diagnostics: bit 0 no X DMA, bit 1 no weight loads, bit 2 no LDS reads / MFMAs
#endif

template <int OFF>  // byte offset encoded in the instruction (16 bits)
__device__ __forceinline__ void lds_read16(u32x4_t& d, unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// waits until at most N of this wave's LDS reads are outstanding; the fragments named here are the ones that have arrived
template <int N>
__device__ __forceinline__ void lds_wait(u32x4_t (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}
// Phase multiply of the stream kernel, unrolled at compile time so that every LDS offset is an instruction immediate.
// Batch B = the four 16-byte pieces of two row tiles (2 ih, 2 ih + 1) of (chunk c, plane pl); two batches of reads are in flight.
template <int PC, int PL>
struct StreamPhase {
    static constexpr int MT = 4, NB = PC * PL * 2;
    template <int B>
    static __device__ __forceinline__ void issue(u32x4_t (&f)[4], unsigned fa0, unsigned fa1) {
        constexpr int c = B / (2 * PL), pl = (B >> 1) % PL, ih = B & 1;
        constexpr int off = ((pl * PC + c) * MT + 2 * ih) * 2048;
        lds_read16<off>(f[0], fa0);
        lds_read16<off>(f[1], fa1);
        lds_read16<off + 2048>(f[2], fa0);
        lds_read16<off + 2048>(f[3], fa1);
    }
    template <int IH>
    static __device__ __forceinline__ void multiply(const BFrag& a, const u32x4_t (&f)[4], f32x4_t (&acc)[MT]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], __builtin_bit_cast(bf16x8_t, f[2 * i]), acc[2 * IH + i], 0, 0, 0);
            acc[2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], __builtin_bit_cast(bf16x8_t, f[2 * i + 1]), acc[2 * IH + i], 0, 0, 0);
        }
    }
    // batches B, B + 1 (fA holds B's pieces, already requested)
    template <int B>
    static __device__ __forceinline__ void pair(const BFrag (&a)[PC], u32x4_t (&fA)[4], u32x4_t (&fB)[4], f32x4_t (&acc)[MT], unsigned fa0, unsigned fa1) {
        if constexpr (B < NB) {
            issue<B + 1>(fB, fa0, fa1);
            lds_wait<4>(fA);
            multiply<0>(a[B / (2 * PL)], fA, acc);
            if constexpr (B + 2 < NB) {
                issue<B + 2>(fA, fa0, fa1);
                lds_wait<4>(fB);
            } else {
                lds_wait<0>(fB);
            }
            multiply<1>(a[B / (2 * PL)], fB, acc);
            pair<B + 2>(a, fA, fB, acc, fa0, fa1);
        }
    }
    // fa0 / fa1: LDS byte addresses of this lane's two pieces in unit 0 of the buffer
    static __device__ __forceinline__ void run(const BFrag (&a)[PC], f32x4_t (&acc)[MT], unsigned fa0, unsigned fa1) {
        u32x4_t fA[4], fB[4];
        issue<0>(fA, fa0, fa1);
        pair<0>(a, fA, fB, acc, fa0, fa1);
    }
};

// The same phase walked by a wave that owns TWO 16-row weight tiles and half of the phase's chunks (PCH of the 2 PCH): every X
// fragment read from LDS feeds four MFMAs instead of two, so a block's fragment traffic is half that of four waves which each
// read the whole tile.  a[jt * PCH + c]: weight fragments of column tile jt, chunk c of this wave's half.
template <int PCH, int PL>
struct StreamPhase2 {
    static constexpr int MT = 4, NB = PCH * PL * 2, PC = 2 * PCH;
    template <int B>
    static __device__ __forceinline__ void issue(u32x4_t (&f)[4], unsigned fa0, unsigned fa1) {
        constexpr int c = B / (2 * PL), pl = (B >> 1) % PL, ih = B & 1;
        constexpr int off = ((pl * PC + c) * MT + 2 * ih) * 2048;
        lds_read16<off>(f[0], fa0);
        lds_read16<off>(f[1], fa1);
        lds_read16<off + 2048>(f[2], fa0);
        lds_read16<off + 2048>(f[3], fa1);
    }
    template <int IH>
    static __device__ __forceinline__ void multiply(const BFrag& a0, const BFrag& a1, const u32x4_t (&f)[4], f32x4_t (&acc)[2][MT]) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            acc[0][2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v[0], __builtin_bit_cast(bf16x8_t, f[2 * i]), acc[0][2 * IH + i], 0, 0, 0);
            acc[1][2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v[0], __builtin_bit_cast(bf16x8_t, f[2 * i]), acc[1][2 * IH + i], 0, 0, 0);
            acc[0][2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v[1], __builtin_bit_cast(bf16x8_t, f[2 * i + 1]), acc[0][2 * IH + i], 0, 0, 0);
            acc[1][2 * IH + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v[1], __builtin_bit_cast(bf16x8_t, f[2 * i + 1]), acc[1][2 * IH + i], 0, 0, 0);
        }
    }
    template <int B>
    static __device__ __forceinline__ void pair(const BFrag (&a)[PC], u32x4_t (&fA)[4], u32x4_t (&fB)[4], f32x4_t (&acc)[2][MT], unsigned fa0, unsigned fa1) {
        if constexpr (B < NB) {
            constexpr int c = B / (2 * PL);
            issue<B + 1>(fB, fa0, fa1);
            lds_wait<4>(fA);
            multiply<0>(a[c], a[PCH + c], fA, acc);
            if constexpr (B + 2 < NB) {
                issue<B + 2>(fA, fa0, fa1);
                lds_wait<4>(fB);
            } else {
                lds_wait<0>(fB);
            }
            multiply<1>(a[c], a[PCH + c], fB, acc);
            pair<B + 2>(a, fA, fB, acc, fa0, fa1);
        }
    }
    static __device__ __forceinline__ void run(const BFrag (&a)[PC], f32x4_t (&acc)[2][MT], unsigned fa0, unsigned fa1) {
        u32x4_t fA[4], fB[4];
        issue<0>(fA, fa0, fa1);
        pair<0>(a, fA, fB, acc, fa0, fa1);
    }
};

constexpr int vmcnt_imm(int n) { return (n & 0xF) | ((n >> 4) << 14) | 0x0F70; }  // s_waitcnt vmcnt(n), other counters untouched

// NP: number of phases, K = NP * PC * 64.  A template constant, so that the phase loop is straight-line code: at a loop
// header the compiler's wait-count bookkeeping drains the whole weight ring.
// SP2: the four compute waves form a 2 x 2 grid - two 32-column halves x two halves of every phase's chunks - instead of four
// 16-column strips over the whole phase (half the LDS fragment reads; the two K halves exchange half of their tiles at the end).
// X goes through a ring of four 32 KB buffers, requested three phases ahead, with ONE barrier per phase: when every wave has
// arrived at the barrier of phase p, the DMA of phase p has landed (the loaders waited for it) and the multiply of phase p - 1
// is over, so its buffer is free for phase p + 3.
template <int MODE, bool LO, int NP, bool SP2>
__global__ __launch_bounds__(512) void gemm_stream_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, int N, int K, int ldx,
                                                          long x_lo, int wpk, FusedEpi e) {
    constexpr int MT = 4, PL = LO ? 2 : 1, PC = LO ? 2 : 4, NBUF = 4;
    constexpr int UB = PL * PC * MT;          // units per buffer (32 KB)
    constexpr int BUF_BYTES = UB * 2048;
    constexpr int DMA_PER_WAVE = UB * 2 / 4;  // DMA instructions a loader wave issues per phase
    constexpr int D = LO ? 10 : 4;            // phases of weight fragments in registers (160 / 128 registers): fetched D - 1 phases ahead
    extern __shared__ uint4 smem[];           // NBUF buffers of UB units, then 4 x 64 partial sums of squares
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = blockIdx.y * 64;
    constexpr int nP = NP;
    float* ssq = reinterpret_cast<float*>(smem + NBUF * UB * 128);

    // Two roles, because a wave's loads complete in order: a wave that waits for its DMA of phase p also waits for every
    // weight load it issued before, which would cap the weight prefetch at one phase (less than the HBM latency).
    if (wave >= 4) {
        // ---- loader waves: the X tiles of phases p + 1 .. p + 3 travel while phase p is multiplied
        const int lw = wave - 4;
        auto dma_phase = [&](int p, int buf) __attribute__((always_inline)) {
#pragma unroll
            for (int k = 0; k < DMA_PER_WAVE; ++k) {
                const int f = lw + 4 * k;            // half unit
                const int u = f >> 1;                // unit (pl, c, i)
                const int pl = u / (PC * MT), c = (u / MT) % PC, i = u % MT;
                if (!(QTTS_STREAM_DBG & 1))
                    dma_half(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)(p * PC + c) * CHUNK, f & 1, smem + (size_t)(buf * UB + u) * 128, lane);
            }
        };
        row_sumsq(e, row0, 64, ssq);
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the sums are in LDS before this wave's first barrier
#pragma unroll
        for (int d = 0; d < NBUF - 1; ++d) dma_phase(min(d, nP - 1), d);
#pragma unroll
        for (int p = 0; p < nP; ++p) {
            __builtin_amdgcn_s_waitcnt(vmcnt_imm((NBUF - 2) * DMA_PER_WAVE));  // this wave's DMA of phase p has landed (two later phases may be under way)
            __builtin_amdgcn_s_barrier();
            dma_phase(min(p + NBUF - 1, nP - 1), (p + NBUF - 1) % NBUF);      // branch-free: past the end the last phase is fetched again into an idle buffer
        }
        __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        __builtin_amdgcn_s_barrier();  // matches the compute waves' barrier in front of the epilogue
        return;
    }
    const int r = lane & 15, g = lane >> 4;
    const unsigned lds0 = (unsigned)reinterpret_cast<size_t>((__attribute__((address_space(3))) char*)smem);
    const bool shared_w = gridDim.y > 1;
    if constexpr (SP2) {
        // ---- compute waves as a 2 x 2 grid: column half ch (weight tiles 2 ch, 2 ch + 1), chunk half kh of every phase
        constexpr int PCH = PC / 2;
        const int ch = wave & 1, kh = wave >> 1;
        const unsigned koff = (unsigned)kh * PCH * MT * 2048;  // this wave's chunks inside a plane of a buffer
        const unsigned fa0 = lds0 + unit_slot(r, g, 0) * 16 + koff, fa1 = lds0 + unit_slot(r, g, 1) * 16 + koff;
        const int n0 = blockIdx.x * 64 + ch * 32;
        const WTileB wt0(W, n0, K, lane, wpk), wt1(W, n0 + 16, K, lane, wpk);
        auto load_w = [&](int p, BFrag (&a)[PC]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < PCH; ++c) {
                const size_t off = (size_t)(p * PC + kh * PCH + c) * wt0.cs;
                if (shared_w) { a[c].load(wt0.p + off, wt0.hs); a[PCH + c].load(wt1.p + off, wt1.hs); }
                else { a[c].load_stream(wt0.p + off, wt0.hs); a[PCH + c].load_stream(wt1.p + off, wt1.hs); }
            }
        };
        f32x4_t acc[2][MT];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        BFrag ring[D][PC];  // every index below is a constant once the phase loop is unrolled
#pragma unroll
        for (int d = 0; d < D - 1; ++d) load_w(min(d, nP - 1), ring[d]);
#pragma unroll
        for (int p = 0; p < nP; ++p) {
            load_w(min(p + D - 1, nP - 1), ring[(p + D - 1) % D]);
            __builtin_amdgcn_s_barrier();
            StreamPhase2<PCH, PL>::run(ring[p % D], acc, fa0 + (p % NBUF) * BUF_BYTES, fa1 + (p % NBUF) * BUF_BYTES);
        }
        __builtin_amdgcn_s_barrier();  // the loader waves' sums of squares are visible, every buffer is idle (the loaders end here)
        // The two K halves of a column half swap tiles: kh keeps row tiles 2 kh, 2 kh + 1 and hands the other two over, then adds
        // what it receives (a + b and b + a are the same number, so both waves add "lower half + upper half").
        f32x4_t* xch = reinterpret_cast<f32x4_t*>(smem);  // [wave][j][2][64]
        f32x4_t keep[2][2];  // (selected with constant indices: a run-time index would put the accumulators in scratch memory)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                xch[((wave * 2 + j) * 2 + i) * 64 + lane] = kh ? acc[j][i] : acc[j][2 + i];
                keep[j][i] = kh ? acc[j][2 + i] : acc[j][i];
            }
        __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        const int other = wave ^ 2;
        f32x4_t fin[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 2; ++i) fin[j][i] = keep[j][i] + xch[((other * 2 + j) * 2 + i) * 64 + lane];
        finished_tile_epilogue<2, MODE>(fin[0], e, ssq + 32 * kh, row0 + 32 * kh, n0, N, r, g);
        finished_tile_epilogue<2, MODE>(fin[1], e, ssq + 32 * kh, row0 + 32 * kh, n0 + 16, N, r, g);
    } else {
        // ---- compute waves: 16 weight rows each, fragments straight to registers D - 1 phases ahead
        const int n0 = blockIdx.x * 64 + wave * 16;
        const unsigned fa0 = lds0 + unit_slot(r, g, 0) * 16, fa1 = lds0 + unit_slot(r, g, 1) * 16;
        const WTileB wt(W, n0, K, lane, wpk);
        auto load_w = [&](int p, BFrag (&a)[PC]) __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < PC; ++c) {
                const bf16_t* src = wt.p + (size_t)(p * PC + c) * wt.cs;
                if (QTTS_STREAM_DBG & 2) src = wt.p;  // one chunk over and over: served by the caches
                if (shared_w) a[c].load(src, wt.hs);
                else a[c].load_stream(src, wt.hs);
            }
        };
        f32x4_t acc[MT];
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        BFrag ring[D][PC];
#pragma unroll
        for (int d = 0; d < D - 1; ++d) load_w(min(d, nP - 1), ring[d]);
#pragma unroll
        for (int p = 0; p < nP; ++p) {
            load_w(min(p + D - 1, nP - 1), ring[(p + D - 1) % D]);
            __builtin_amdgcn_s_barrier();
            if (!(QTTS_STREAM_DBG & 4)) StreamPhase<PC, PL>::run(ring[p % D], acc, fa0 + (p % NBUF) * BUF_BYTES, fa1 + (p % NBUF) * BUF_BYTES);
            else acc[0] += __builtin_bit_cast(f32x4_t, ring[p % D][0].v[0]) + __builtin_bit_cast(f32x4_t, ring[p % D][PC - 1].v[1]);
        }
        __builtin_amdgcn_s_barrier();  // the loader waves' partial sums of squares are visible
        finished_tile_epilogue<MT, MODE>(acc, e, ssq, row0, n0, N, r, g);
    }
}

// ------------------------------------------------------------------------------------------------ k_split
constexpr int KS_WAVES = 8;   // waves of a k_split block
constexpr int KS_GROUP = 6;   // most chunks a wave requests at once (X units in LDS, weight fragments in registers); half with a remainder plane

// MT: 16-row token tiles a block owns (every weight fragment is multiplied with MT X fragments: the X tile of a tile row is read
// by N / 16 column blocks either way, the weights by rows / (16 MT) row slices)
template <bool LO, int MT = 1>
__global__ __launch_bounds__(64 * KS_WAVES) void gemm_k_split_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ X, int N, int K,
                                                                     int ldx, long x_lo, int wpk, int gcap, FusedEpi e) {
    constexpr int NW = KS_WAVES, PL = LO ? 2 : 1, G = KS_GROUP / PL;
    QTTS_TS(0);
    extern __shared__ uint4 ksm[];  // red[NW][MT][64] partial tiles, then every wave's own X units [NW][PL][gcap][MT][128]
    f32x4_t* red = reinterpret_cast<f32x4_t*>(ksm);
    uint4* xw = ksm + NW * MT * 64 + (size_t)(threadIdx.x >> 6) * PL * gcap * MT * 128;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int n0 = tile * 16;
    const int row0 = blockIdx.y * 16 * MT;
    const int nC = K / CHUNK;
    const int ca = wave * nC / NW, cb = (wave + 1) * nC / NW;
    const WTileB wt(W, n0, K, lane, wpk);
    const bool shared_w = gridDim.y > 1;
    const int s0 = unit_slot(r, g, 0), s1 = unit_slot(r, g, 1);
    f32x4_t acc[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // wave i < MT finishes row tile i; what its epilogue reads (bias, residual, norm weight) is requested now, so that the
    // memory round trip runs beside the K loop instead of after it
    const int em = row0 + 16 * (wave < MT ? wave : 0) + (lane & 15), en = n0 + 4 * (lane >> 4);
    float4 pre_b = make_float4(0.f, 0.f, 0.f, 0.f), pre_r = pre_b, pre_w = pre_b;
    if (wave < MT) {
        if (e.bias) pre_b = *reinterpret_cast<const float4*>(e.bias + en);
        if (e.resid) pre_r = *reinterpret_cast<const float4*>(e.resid + (size_t)em * N + en);
        pre_w = *reinterpret_cast<const float4*>(e.nw + en);
    }

    for (int c0 = ca; c0 < cb; c0 += gcap) {
        if (c0 != ca) __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the previous group's LDS reads are done before its units are overwritten
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (u < gcap && c0 + u < cb) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        dma_half(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)(c0 + u) * CHUNK, 0, xw + ((pl * gcap + u) * MT + i) * 128, lane);
                        dma_half(X + (pl ? x_lo : 0), row0 + 16 * i, ldx, (long)(c0 + u) * CHUNK, 1, xw + ((pl * gcap + u) * MT + i) * 128, lane);
                    }
            }
        }
        BFrag a[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (u < gcap && c0 + u < cb) {
                if (shared_w) a[u].load(wt.p + (size_t)(c0 + u) * wt.cs, wt.hs);
                else a[u].load_stream(wt.p + (size_t)(c0 + u) * wt.cs, wt.hs);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the wave's own DMA has landed (its units are private, no barrier)
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (u < gcap && c0 + u < cb) {
#pragma unroll
                for (int pl = 0; pl < PL; ++pl)
#pragma unroll
                    for (int i = 0; i < MT; ++i) acc[i] = mma(a[u], unit_frag(xw + ((pl * gcap + u) * MT + i) * 128, s0, s1), acc[i]);
            }
        }
    }
    QTTS_TS(1);
#pragma unroll
    for (int i = 0; i < MT; ++i) red[(wave * MT + i) * 64 + lane] = acc[i];
    __syncthreads();
    QTTS_TS(2);
    // lane l of wave i < MT: row 16 i + (l & 15), columns n0 + 4 (l >> 4) .. + 3; partial tiles are added in wave order
    if (wave < MT) {
        const int l = lane;
        f32x4_t v = red[wave * 64 + l];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[(w * MT + wave) * 64 + l];
        const int m = em, n = en;
        float x[4] = {v[0], v[1], v[2], v[3]};
        if (e.bias) {
            x[0] += pre_b.x; x[1] += pre_b.y; x[2] += pre_b.z; x[3] += pre_b.w;
        }
        if (e.resid) {
            x[0] = pre_r.x + x[0]; x[1] = pre_r.y + x[1]; x[2] = pre_r.z + x[2]; x[3] = pre_r.w + x[3];
        }
        *reinterpret_cast<float4*>(e.x_out + (size_t)m * N + n) = make_float4(x[0], x[1], x[2], x[3]);
        const float4 w4 = pre_w;
        bf16_t* xb = (bf16_t*)e.xb_out + (size_t)m * N + n;
        store_split<bf16_t>(xb, e.xb_lo, x[0] * w4.x);
        store_split<bf16_t>(xb + 1, e.xb_lo, x[1] * w4.y);
        store_split<bf16_t>(xb + 2, e.xb_lo, x[2] * w4.z);
        store_split<bf16_t>(xb + 3, e.xb_lo, x[3] * w4.w);
        // the row's sum of squares over this block's 16 columns: the four lanes 16 apart hold them
        float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        const auto add = [](float x0, float x1) { return x0 + x1; };
        ss = fold_rows(ss, add);
        ss = fold_halves(ss, add);
        if ((l >> 4) == 0) e.ss_out[(long)tile * e.ss_ld + m] = ss;
    }
    QTTS_TS(3);
}

}  // namespace

bool fused_gemm_supported(int dtype, int hidden) { return dtype == BF16 && hidden % 128 == 0 && hidden <= 2048; }

int fused_rows(int M) { return M <= 16 ? 16 : (M <= 32 ? 32 : (M + 63) / 64 * 64); }

// more than 64 KB of dynamic LDS needs the limit raised once per kernel (per device: the attribute lives with the loaded code object)
static void raise_lds_limit(const void* fn, int slot) {
    static std::mutex mu;
    static bool raised[64][64] = {};
    int dev = 0;
    QTTS_HIP(hipGetDevice(&dev));
    QTTS_CHECK(dev >= 0 && dev < 64 && slot >= 0 && slot < 64, ERR_STATE, "fused gemm: device ordinal out of range");
    std::lock_guard<std::mutex> lock(mu);
    if (!raised[dev][slot]) {
        QTTS_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        raised[dev][slot] = true;
    }
}

// K slices taken by separate waves of a block (4 waves per slice).  Measured at 64 rows, K = 1024 (us per launch, one / two
// slices, stand-alone): gate|up N = 6144 8.95 / 7.26, q|k|v N = 4096 6.07 / 5.64, a 2048-wide head 5.83 / 5.38; inside the decode
// frame the wide launch keeps its gain (9.46 / 8.57) and the narrower ones lose it (6.67 / 7.00), so two slices are taken from
// N = 6144 on.  Four slices (1024 threads: the weight fragments no longer fit the registers) 12.7 and worse.
// QTTS_FULL_K_SLICES: 1 = one slice whenever K <= 1024, 2 = two slices whenever K >= 1024.
static int full_k_slices(int K, int N) {
    static const int forced = getenv("QTTS_FULL_K_SLICES") ? atoi(getenv("QTTS_FULL_K_SLICES")) : 0;
    if (K % (2 * CHUNK) != 0) return 1;
    if (K > 1024) return 2;  // a wave holds at most 16 chunks of weights
    if (K < 1024 || forced == 1) return 1;
    return forced == 2 || N >= 6144 ? 2 : 1;
}
constexpr size_t FULL_K_LDS_MAX = 150 * 1024;  // X tile + partial sums + hand-over tiles of one block
static size_t full_k_lds(int K, int mt, int planes, int kh) {
    return (size_t)planes * (K / CHUNK) * mt * 2048 + 4 * 64 * 4 + (size_t)(kh - 1) * 4 * mt * 64 * 16;
}

template <int MT, int MODE, int KH, bool LO>
static void launch_full_k_inst(hipStream_t st, const bf16_t* W, int wpk, const bf16_t* X, long x_lo, int rows, int N, int K, int ldx, const FusedEpi& e) {
    const size_t lds = full_k_lds(K, MT, LO ? 2 : 1, KH);
    dim3 grid(N / 64, rows / (16 * MT));
    if (lds > 64 * 1024)
        raise_lds_limit((const void*)gemm_full_k_kernel<MT, MODE, KH, LO>, ((MT == 4 ? 0 : (MT == 2 ? 1 : 2)) * 2 + MODE) * 4 + (KH - 1) * 2 + (LO ? 1 : 0));
    hipLaunchKernelGGL((gemm_full_k_kernel<MT, MODE, KH, LO>), grid, dim3(256 * KH), lds, st, W, X, N, K, ldx, x_lo, wpk, e);
}
template <int MT, int MODE>
static void launch_full_k_mode(hipStream_t st, const bf16_t* W, int wpk, const bf16_t* X, long x_lo, int rows, int N, int K, int ldx, const FusedEpi& e) {
    const int kh = full_k_slices(K, N);
#define QTTS_FULL_K_KH(KHV)                                                                          \
    if (x_lo) launch_full_k_inst<MT, MODE, KHV, true>(st, W, wpk, X, x_lo, rows, N, K, ldx, e);      \
    else launch_full_k_inst<MT, MODE, KHV, false>(st, W, wpk, X, 0, rows, N, K, ldx, e)
    if (kh == 2) { QTTS_FULL_K_KH(2); }
    else { QTTS_FULL_K_KH(1); }
#undef QTTS_FULL_K_KH
}

bool full_k_fits(int K, bool lo) { return K % CHUNK == 0 && K <= 2048 && (K <= 1024 || K % (2 * CHUNK) == 0) && full_k_lds(K, 1, lo ? 2 : 1, 2) <= FULL_K_LDS_MAX; }

void launch_gemm_full_k(hipStream_t st, WMat Wm, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e) {
    QTTS_CHECK(M >= 1 && N % 64 == 0 && full_k_fits(K, x_lo != 0), ERR_SHAPE, "gemm_full_k: N must be a multiple of 64 and a 16-row X tile must fit LDS");
    QTTS_CHECK(e.mode == FUSED_PLAIN || e.mode == FUSED_SWIGLU, ERR_BAD_ARG, "gemm_full_k: bad epilogue");
    QTTS_CHECK(e.mode != FUSED_PLAIN || (e.out_f32 && e.ld_out >= N), ERR_BAD_ARG, "gemm_full_k: missing output");
    QTTS_CHECK(e.mode != FUSED_SWIGLU || e.act, ERR_BAD_ARG, "gemm_full_k: missing activation output");
    QTTS_CHECK(!e.ss_in || (e.ntiles_in >= 1 && e.H_in >= 1), ERR_BAD_ARG, "gemm_full_k: bad norm partials");
    const int rows = fused_rows(M);
    const bf16_t* w = (const bf16_t*)Wm.p;
    const bf16_t* x = (const bf16_t*)X;
    const int wpk = Wm.packed ? 1 : 0;
    const int planes = x_lo ? 2 : 1, kh = full_k_slices(K, N);
    // row tiles per block: the most that fits LDS and still gives about 192 blocks (a 64-row launch of 64 column blocks
    // becomes 256 blocks of 16 rows)
    int mt = rows >= 64 ? 4 : rows / 16;
    static const long min_blocks = getenv("QTTS_FULL_K_MIN_BLOCKS") ? atol(getenv("QTTS_FULL_K_MIN_BLOCKS")) : 192;
    while (mt > 1 && (full_k_lds(K, mt, planes, kh) > FULL_K_LDS_MAX || (long)(N / 64) * (rows / (16 * mt)) < min_blocks)) mt >>= 1;
#define QTTS_FULL_K(MTV)                                                                                        \
    if (e.mode == FUSED_PLAIN) launch_full_k_mode<MTV, FUSED_PLAIN>(st, w, wpk, x, x_lo, rows, N, K, ldx, e);   \
    else launch_full_k_mode<MTV, FUSED_SWIGLU>(st, w, wpk, x, x_lo, rows, N, K, ldx, e)
    if (mt == 1) { QTTS_FULL_K(1); }
    else if (mt == 2) { QTTS_FULL_K(2); }
    else { QTTS_FULL_K(4); }
#undef QTTS_FULL_K
}

bool gemm_stream_fits(int M, int N, int K, bool lo) {
    return fused_rows(M) % 64 == 0 && N % 64 == 0 && (K == 1024 || K == 2048);  // the instantiated phase counts
}
// Measured at 64 rows (us per launch, weights streamed from HBM): N = 12288, K = 2048, two planes: 17.9 against 34.6 for the
// whole-K tile kernel and 23.8 + 5.3 for split-K slabs + the SwiGLU pass; N = 4096: 12.8 against 10.4 (64 blocks are too few);
// K = 1024: the whole-K kernel wins at every N.  QTTS_GEMM_STREAM: 0 never, 1 (default) by this rule, 2 whenever it fits.
bool gemm_stream_preferred(int M, int N, int K, bool lo, bool shared) {
    static const int mode = getenv("QTTS_GEMM_STREAM") ? atoi(getenv("QTTS_GEMM_STREAM")) : 1;
    if (mode == 0 || !gemm_stream_fits(M, N, K, lo)) return false;
    if (mode >= 2) return true;
    // several streams in flight: wherever it fits (in the whole job that beats keeping the whole-K tile for the code predictor's
    // q|k|v and heads, 2000 against 1954 audio-s/s without the codec)
    if (shared) return true;
    return K == 2048 && N >= 8192;
}

int tuning_forced() {
    static const int forced = [] {
        const char* v = getenv("QTTS_TUNE");
        if (!v) return (int)TUNE_UNSET;
        return std::string(v) == "shared" ? (int)TUNE_SHARED : (std::string(v) == "alone" ? (int)TUNE_ALONE : (int)TUNE_UNSET);
    }();
    return forced;
}

template <int MODE, bool LO, int NP>
static void launch_stream_np(hipStream_t st, const bf16_t* W, int wpk, const bf16_t* X, long x_lo, int rows, int N, int K, int ldx, const FusedEpi& e, int slot) {
    // QTTS_STREAM_GRID=1: compute waves as a 2 x 2 grid.  Same time as four strips alone on the GPU (17.8 against 17.9 us at
    // N = 12288, 4.80 ms per frame either way) and no better with three sessions in flight (1799 against 1825 audio-s/s): the kernel
    // is bound on its load side, not by the fragment reads the grid halves - the strips stay the default.
    static const bool strips = !(getenv("QTTS_STREAM_GRID") && atoi(getenv("QTTS_STREAM_GRID")) == 1);
    const size_t lds = (size_t)2 * 64 * 1024 + 4 * 64 * 4;
    if (strips) {
        raise_lds_limit((const void*)gemm_stream_kernel<MODE, LO, NP, false>, slot);
        hipLaunchKernelGGL((gemm_stream_kernel<MODE, LO, NP, false>), dim3(N / 64, rows / 64), dim3(512), lds, st, W, X, N, K, ldx, x_lo, wpk, e);
    } else {
        raise_lds_limit((const void*)gemm_stream_kernel<MODE, LO, NP, true>, slot + 8);
        hipLaunchKernelGGL((gemm_stream_kernel<MODE, LO, NP, true>), dim3(N / 64, rows / 64), dim3(512), lds, st, W, X, N, K, ldx, x_lo, wpk, e);
    }
}
template <int MODE, bool LO>
static void launch_stream_inst(hipStream_t st, const bf16_t* W, int wpk, const bf16_t* X, long x_lo, int rows, int N, int K, int ldx, const FusedEpi& e) {
    constexpr int PHASE = CHUNK * (LO ? 2 : 4);
    const int slot = 40 + MODE * 4 + (LO ? 2 : 0);
    if (K == 2048) launch_stream_np<MODE, LO, 2048 / PHASE>(st, W, wpk, X, x_lo, rows, N, K, ldx, e, slot);
    else launch_stream_np<MODE, LO, 1024 / PHASE>(st, W, wpk, X, x_lo, rows, N, K, ldx, e, slot + 1);
}

void launch_gemm_stream(hipStream_t st, WMat Wm, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e) {
    QTTS_CHECK(M >= 1 && gemm_stream_fits(M, N, K, x_lo != 0), ERR_SHAPE, "gemm_stream: 64-row multiples, N % 64 == 0 and K a multiple of the phase length");
    QTTS_CHECK(e.mode == FUSED_PLAIN || e.mode == FUSED_SWIGLU, ERR_BAD_ARG, "gemm_stream: bad epilogue");
    QTTS_CHECK(e.mode != FUSED_PLAIN || (e.out_f32 && e.ld_out >= N), ERR_BAD_ARG, "gemm_stream: missing output");
    QTTS_CHECK(e.mode != FUSED_SWIGLU || e.act, ERR_BAD_ARG, "gemm_stream: missing activation output");
    const int rows = fused_rows(M);
    const bf16_t* w = (const bf16_t*)Wm.p;
    const bf16_t* x = (const bf16_t*)X;
    const int wpk = Wm.packed ? 1 : 0;
    if (e.mode == FUSED_PLAIN) {
        if (x_lo) launch_stream_inst<FUSED_PLAIN, true>(st, w, wpk, x, x_lo, rows, N, K, ldx, e);
        else launch_stream_inst<FUSED_PLAIN, false>(st, w, wpk, x, 0, rows, N, K, ldx, e);
    } else {
        if (x_lo) launch_stream_inst<FUSED_SWIGLU, true>(st, w, wpk, x, x_lo, rows, N, K, ldx, e);
        else launch_stream_inst<FUSED_SWIGLU, false>(st, w, wpk, x, 0, rows, N, K, ldx, e);
    }
}

int fused_resid_tiles(int N) { return N / 16; }

void launch_gemm_k_split(hipStream_t st, WMat Wm, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e, bool shared) {
    QTTS_CHECK(M >= 1 && N % 16 == 0 && K % CHUNK == 0, ERR_SHAPE, "gemm_k_split: N must be a multiple of 16 and K of 64");
    QTTS_CHECK(e.mode == FUSED_RESID && e.x_out && e.xb_out && e.nw && e.ss_out && e.ss_ld >= fused_rows(M), ERR_BAD_ARG,
               "gemm_k_split: missing residual outputs");
    const int rows = fused_rows(M);
    const bf16_t* w = (const bf16_t*)Wm.p;
    const bf16_t* x = (const bf16_t*)X;
    const int wpk = Wm.packed ? 1 : 0;
    const int planes = x_lo ? 2 : 1;
    const int per_wave = (K / CHUNK + KS_WAVES - 1) / KS_WAVES;          // chunks of the longest wave slice
    // X chunks a wave requests at once: six alone on the GPU, three with several decode streams in flight (56 instead of 104 KB of
    // LDS per block: blocks of two streams share a CU)
    static const int group_env = getenv("QTTS_KS_GROUP") ? atoi(getenv("QTTS_KS_GROUP")) : 0;
    // token rows per block: 16 alone on the GPU, 32 with several decode streams in flight (every weight fragment meets two X
    // fragments: half the weight reads through the L1s and half the blocks; 4.82 against 4.69 ms per frame alone, 1953 against
    // 1924 audio-s/s without the codec with three sessions).  QTTS_KS_ROWS=16 / 32 forces either.
    static const int rows_env = getenv("QTTS_KS_ROWS") ? atoi(getenv("QTTS_KS_ROWS")) : 0;
    const int want = rows_env == 16 || rows_env == 32 ? rows_env : (shared ? 32 : 16);
    const int mt = (want == 32 && rows % 32 == 0 && !x_lo) ? 2 : 1;
    const int group = std::max(planes, std::min(KS_GROUP, group_env > 0 ? group_env : (shared ? 3 : KS_GROUP)));
    int gcap = std::max(1, std::min(group / planes, per_wave));
    if (mt == 2) gcap = std::min(gcap, 3);  // 112 KB of LDS at most
    dim3 grid(N / 16, rows / (16 * mt));
    const size_t lds = (size_t)KS_WAVES * mt * 64 * 16 + (size_t)KS_WAVES * planes * gcap * mt * 2048;
    if (x_lo) {
        if (lds > 64 * 1024) raise_lds_limit((const void*)gemm_k_split_kernel<true>, 60);
        hipLaunchKernelGGL((gemm_k_split_kernel<true>), grid, dim3(64 * KS_WAVES), lds, st, w, x, N, K, ldx, x_lo, wpk, gcap, e);
    } else if (mt == 2) {
        if (lds > 64 * 1024) raise_lds_limit((const void*)gemm_k_split_kernel<false, 2>, 62);
        hipLaunchKernelGGL((gemm_k_split_kernel<false, 2>), grid, dim3(64 * KS_WAVES), lds, st, w, x, N, K, ldx, 0L, wpk, gcap, e);
    } else {
        if (lds > 64 * 1024) raise_lds_limit((const void*)gemm_k_split_kernel<false>, 61);
        hipLaunchKernelGGL((gemm_k_split_kernel<false>), grid, dim3(64 * KS_WAVES), lds, st, w, x, N, K, ldx, 0L, wpk, gcap, e);
    }
}

}  // namespace qtts
