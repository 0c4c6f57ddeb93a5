// Causal / transposed convolutions of the codec vocoder as implicit GEMM with the input tile held in LDS
// (qwen_tts/src/audio/tokenizer/v2/causal_conv.rs:15-176, audio/decoder/residual_unit.rs:11-96).
//
// With channels-last strips the taps of a conv are row offsets, so a block that owns 16*MT consecutive output rows
// needs input rows [m0, m0 + 16*MT + (taps-1)*dil): they are loaded ONCE into LDS (16-byte vectors, XOR-swizzled so that
// the MFMA B-fragment reads of 16 consecutive rows hit distinct banks) and every tap reads them at its own row offset.
// After that single barrier the K loop only streams weight fragments (L2-resident, four chunks in flight per wave) into
// MFMAs; the element-wise tail of the layer (conv_epi.hpp) runs on the accumulators after a transpose through LDS that makes
// its global accesses contiguous.  Compared with the generic
// tap-addressed GEMM this removes the 7-fold re-read of the input and the per-phase barriers.
#include "conv_epi.hpp"
#include <algorithm>
#include <cstdlib>

#ifndef QTTS_W_STEP
#define QTTS_W_STEP(s) (s)  // (development probes redefine it to make every step read the same weight fragments)
#endif

namespace qtts {

namespace {

constexpr int LDS_CEILING = 152 * 1024;  // dynamic LDS any kernel here may ask for (160 KB per CU on gfx950, minus the static arrays)

template <typename T>
struct CFrag;
template <>
struct CFrag<bf16_t> {
    static constexpr int CHUNK = 64, EPL = 16;
    bf16x8_t v[2];
    __device__ __forceinline__ void load(const bf16_t* p, int hs) {  // hs: elements between the lane's two 16-byte pieces
        v[0] = *reinterpret_cast<const bf16x8_t*>(p);
        v[1] = *reinterpret_cast<const bf16x8_t*>(p + hs);
    }
    static __device__ __forceinline__ f32x4_t mma(const CFrag& a, const CFrag& b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], b.v[1], c, 0, 0, 0);
        return c;
    }
    // one 16-byte half of the B fragment at a time (the halves of a chunk in order: the same sums as mma)
    static constexpr int HALF_MFMAS = 1;
    static __device__ __forceinline__ f32x4_t mma_half(const CFrag& a, int half, const uint4& h, f32x4_t c) {
        bf16x8_t b;
        __builtin_memcpy(&b, &h, 16);
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[half], b, c, 0, 0, 0);
    }
};
template <>
struct CFrag<float> {
    static constexpr int CHUNK = 32, EPL = 8;
    float v[8];
    __device__ __forceinline__ void load(const float* p, int hs) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + hs);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ f32x4_t mma(const CFrag& a, const CFrag& b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
        return c;
    }
    static constexpr int HALF_MFMAS = 4;
    static __device__ __forceinline__ f32x4_t mma_half(const CFrag& a, int half, const uint4& h, f32x4_t c) {
        float b[4];
        __builtin_memcpy(b, &h, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[4 * half + j], b[j], c, 0, 0, 0);
        return c;
    }
};

// LDS image of an activation tile: rows of vpr 16-byte slots (vpr a multiple of 8), read as MFMA B fragments - lane (r, g) of a
// 16-row fragment takes slots 2g and 2g + 1 of a row's 64-element chunk.  The LDS serves a 16-byte read of a wave in four groups
// of 16 lanes, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32: eight rows of one g and eight rows of the next, and
// the group takes one cycle when its 16 slots cover all 64 banks.  Two measures give that for every tap offset:
//  * frag_row: the lanes of a fragment take the 16 rows in an order that puts the even rows into one half of each group and the
//    odd rows into the other (the epilogues use the same order when they put the accumulators back into rows);
//  * tile_pcol: inside aligned groups of 16 slots (8 where a row is an odd number of 128-byte halves, which then alternate
//    between the two halves of the banks) a row's slot pairs are rotated by half the row number and the two slots of a pair are
//    exchanged in odd rows, so the eight rows of a half land on eight different pairs and the two halves on different slots.
// (With the plain XOR swizzle slot = col ^ (row & 7) every such read took two cycles per group.)
__device__ __forceinline__ int frag_row(int lane) {
    const int q = (lane + 4) & 15;
    return ((q & 7) << 1) | (q >> 3);
}
__device__ __forceinline__ int tile_pcol(int row, int col, int vpr) {  // slot of column `col` (in 16-byte units) inside its row
    const int m = (vpr & 15) ? 7 : 15, k = (vpr & 15) ? row >> 1 : row;
    return (col & ~m) | ((((col >> 1) + (k >> 1)) & (m >> 1)) << 1) | ((col ^ k) & 1);
}
__device__ __forceinline__ int tile_col_at(int row, int pcol, int vpr) {  // the column that slot `pcol` of a row holds
    const int m = (vpr & 15) ? 7 : 15, k = (vpr & 15) ? row >> 1 : row;
    return (pcol & ~m) | ((((pcol >> 1) - (k >> 1)) & (m >> 1)) << 1) | ((pcol ^ k) & 1);
}

// Input tile -> LDS by DMA (no registers, every load of the tile in flight at once).  A wave instruction fills 64
// consecutive 16-byte slots, so the swizzle is applied on the source side: slot (row, pcol) takes column tile_col_at(row, pcol).
// The last instruction may run up to 63 slots past `total`: the tile allocation is rounded up to 1 KiB and the rows it
// reads lie inside the slack every caller keeps readable behind the activations.
template <typename T>
__device__ __forceinline__ void tile_load_dma(uint4* tile, const T* src, int C, int vpr, int total) {
    constexpr int VPE = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nblk = (total + 63) >> 6;
    for (int k = wave; k < nblk; k += 4) {
        const int L = k * 64 + lane;
        const int row = L / vpr, pcol = L - row * vpr;
        const T* p = src + (long)row * C + tile_col_at(row, pcol, vpr) * VPE;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(tile + k * 64), 16, 0, 0);
    }
}

// The same tile from the f32 stream the previous layer left: every 16-byte slot is eight channels read as f32, put through
// SnakeBeta (alpha / den of the consuming conv's input activation, as snake_prep leaves them in sn[0 .. C) / sn[C .. 2C) in LDS)
// and rounded to bf16 - bit for bit the activated copy a producer would have stored, without its round trip through memory.
// Slots are taken four at a time so that eight loads per thread are in flight.
__device__ __forceinline__ void tile_load_snake(uint4* tile, const float* src, int C, int vpr, int total, const float* sn) {
    constexpr int U = 4;
    for (int L0 = threadIdx.x; L0 < total; L0 += 256 * U) {
        float4 a[U], b[U];
        int col[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int L = L0 + 256 * k;
            const int Lc = L < total ? L : L0;
            const int row = Lc / vpr, pcol = Lc - row * vpr;
            col[k] = tile_col_at(row, pcol, vpr) * 8;
            const float* p = src + (long)row * C + col[k];
            a[k] = *reinterpret_cast<const float4*>(p);
            b[k] = *reinterpret_cast<const float4*>(p + 4);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int L = L0 + 256 * k;
            if (L >= total) continue;
            const float x[8] = {a[k].x, a[k].y, a[k].z, a[k].w, b[k].x, b[k].y, b[k].z, b[k].w};
            float al[8], dn[8];
            *reinterpret_cast<float4*>(al) = *reinterpret_cast<const float4*>(sn + col[k]);
            *reinterpret_cast<float4*>(al + 4) = *reinterpret_cast<const float4*>(sn + col[k] + 4);
            *reinterpret_cast<float4*>(dn) = *reinterpret_cast<const float4*>(sn + C + col[k]);
            *reinterpret_cast<float4*>(dn + 4) = *reinterpret_cast<const float4*>(sn + C + col[k] + 4);
            u16 h[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) h[q] = f32_to_bf16_bits(snake_apply<bf16_t>(x[q], al[q], dn[q]));
            tile[L] = make_uint4((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16),
                                 (unsigned)h[4] | ((unsigned)h[5] << 16), (unsigned)h[6] | ((unsigned)h[7] << 16));
        }
    }
}

// Block -> tile map of the conv kernels.  The grid is one-dimensional and a multiple of 8 long; blocks are handed to the eight
// XCDs round-robin, so block b works on tile p = (b % 8) * (grid / 8) + b / 8: consecutive tiles - the column blocks of one row
// block, then the next row block - run on ONE XCD and find the input rows they share (the same tile for every column block, the
// halo of a neighbouring row block) in its L2 instead of fetching them once per XCD.  Placement is a matter of speed only: any
// other hand-out of the blocks computes the same result.  xcd == 0: tile p = b.  False: the block has no tile.
__device__ __forceinline__ bool conv_block_tile(long total, int ncol, int xcd, int& colb, long& rowb) {
    const long b = blockIdx.x;
    const long p = xcd ? (b & 7) * ((long)gridDim.x >> 3) + (b >> 3) : b;
    if (p >= total) return false;
    rowb = p / ncol;
    colb = (int)(p - rowb * ncol);
    return true;
}
// grid length for `total` tiles (false: too many for one launch)
static bool conv_grid(long total, int xcd, unsigned& grid) {
    const long g = xcd ? (total + 7) / 8 * 8 : total;
    if (g < 1 || g > 0x7ffffff8L) return false;
    grid = (unsigned)g;
    return true;
}
static int conv_xcd_order() {  // QTTS_CONV_XCD=0: tiles in launch order
    static const int on = getenv("QTTS_CONV_XCD") ? (atoi(getenv("QTTS_CONV_XCD")) != 0) : 1;
    return on;
}

// One epilogue pass over the transposed accumulators in LDS: RT rows x 4 * GPR columns starting at column n0.  A thread keeps
// the same four columns for every row it visits, so the column parameters are loaded once, and the residual loads of
// up to eight rows are issued before the first of them is needed.
template <typename T, int RT, int GPR, int OP>
__device__ __forceinline__ void epi_pass(const ConvEpi& e, const float* otile, const int* row_u, const int* row_tp, int n0) {
    static_assert(256 % GPR == 0 && RT % (256 / GPR) == 0, "epilogue pass geometry");
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    // f32 runs call the full-precision sine: unrolling eight of those tails spills, and their speed is not the point
    constexpr int U = sizeof(T) == 4 ? 1 : (IT >= 8 ? 8 : IT);
    static_assert(IT % U == 0, "epilogue pass geometry");
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
    EpiCols pc;
    conv_epi_cols<T>(e, n0 + 4 * c4, pc);
    for (int it0 = 0; it0 < IT; it0 += U) {
        float rr[U][4];
        long off[U];
        int tp[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const int row = (it0 + k) * RPI + r0;
            tp[k] = row_tp[row];
            off[k] = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[k][q] = 0.f;
            if (tp[k] >= 0) {
                off[k] = conv_epi_offset(e, pc, row_u[row], tp[k]);
                conv_epi_resid<T>(e, off[k], rr[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (tp[k] < 0) continue;
            const int row = (it0 + k) * RPI + r0;
            const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            conv_epi_finish<T>(e, pc, row_u[row], off[k], v, rr[k]);
        }
    }
}

// The vocoder's usual tail as a pass of its own - bias, optional f32 residual, f32 stream out and / or the SnakeBeta copy in the
// model dtype (conv_epi_stream_ok): the uniform choices are made once per pass instead of once per four values, a row's output
// offset comes ready from LDS (row_off: element offset of the row's first output row, < 0 = no output), and the instruction
// stream is a fraction of the general tail's (whose eight unrolled copies with every activation inlined do not fit the
// instruction cache).  Values and stores are the general tail's.
template <typename T, int RT, int GPR, int OP>
__device__ __forceinline__ void epi_pass_stream(const ConvEpi& e, const float* otile, const long* row_off, int n0) {
    static_assert(256 % GPR == 0 && RT % (256 / GPR) == 0, "epilogue pass geometry");
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    constexpr int U = sizeof(T) == 4 ? 1 : (IT >= 8 ? 8 : IT);
    static_assert(IT % U == 0, "epilogue pass geometry");
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
    EpiCols pc;
    conv_epi_cols<T>(e, n0 + 4 * c4, pc);
    if (e.Cvalid && pc.co >= e.Cvalid) return;  // columns past the layer's real width are padding
    const long coff = (long)pc.j * e.ld_out + pc.co;
    const float* rs = e.resid_f32 ? e.resid_f32 + coff : nullptr;
    float* of = e.out_f32 ? e.out_f32 + coff : nullptr;
    T* o2 = e.out2 ? (T*)e.out2 + coff : nullptr;
    const long lo2 = e.out2_lo;
    const bool exact = e.out_f32 || e.resid_f32 || e.out_lo || lo2;  // SnakeBeta of the f32 value (else of the value rounded to the model dtype)
    for (int it0 = 0; it0 < IT; it0 += U) {
        float rr[U][4];
        long off[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            off[k] = row_off[(it0 + k) * RPI + r0];
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[k][q] = 0.f;
            if (rs && off[k] >= 0) ld4(rs + off[k], rr[k]);
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (off[k] < 0) continue;
            const float4 a4 = *reinterpret_cast<const float4*>(otile + ((it0 + k) * RPI + r0) * OP + 4 * c4);
            float v[4] = {a4.x + pc.bias[0], a4.y + pc.bias[1], a4.z + pc.bias[2], a4.w + pc.bias[3]};
            if (rs) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = rr[k][q] + v[q];
            }
            if (of) *reinterpret_cast<float4*>(of + off[k]) = make_float4(v[0], v[1], v[2], v[3]);
            if (o2) {
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) y[q] = snake_apply<T>(exact ? v[q] : to_f32<T>(from_f32<T>(v[q])), pc.al[q], pc.den[q]);
                store4_split<T>(o2 + off[k], lo2, y);
            }
        }
    }
}
static bool conv_epi_stream_ok(const ConvEpi& e) {  // QTTS_EPI_STREAM=0: every conv through the general tail
    static const bool off = getenv("QTTS_EPI_STREAM") && atoi(getenv("QTTS_EPI_STREAM")) == 0;
    return !off && e.bias && !e.bias_u && !e.scale && e.act == 0 && !e.resid && !e.out && (e.out_f32 || e.out2) && (!e.out2 || e.act2 == 0);
}
// element offset of the first output row of every tile row (the general tail's conv_epi_offset without its column part)
template <int RT>
__device__ __forceinline__ void epi_row_offsets(const ConvEpi& e, long m0, long* row_off) {
    for (int t = threadIdx.x; t < RT; t += 256) {
        const long m = m0 + t;
        long o = -1;
        if (m < e.M) {
            const long u = m / e.S_in;
            const int tp = (int)(m - u * e.S_in);
            if (tp < e.lens[u]) o = (u * e.S_out + e.P_out + (long)tp * e.r) * e.ld_out;
        }
        row_off[t] = o;
    }
}

// The same pass with the residual rows already in registers (res[k] = the four values of row k * RPI + r0 the thread read when
// it built the input tile): nothing is loaded from the residual stream.
template <typename T, int RT, int GPR, int OP>
__device__ __forceinline__ void epi_pass_regs(const ConvEpi& e, const float* otile, const int* row_u, const int* row_tp, int n0,
                                              const float (&res)[RT / (256 / GPR)][4]) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
    EpiCols pc;
    conv_epi_cols<T>(e, n0 + 4 * c4, pc);
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int row = k * RPI + r0;
        const int tp = row_tp[row];
        if (tp < 0) continue;
        const long off = conv_epi_offset(e, pc, row_u[row], tp);
        const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
        float v[4] = {a4.x, a4.y, a4.z, a4.w};
        conv_epi_finish<T>(e, pc, row_u[row], off, v, res[k]);
    }
}

// Rows [row0, row0 + RT) of an f32-stream input tile, loaded in the epilogue's thread layout (thread = four channels of every
// RPI-th row, NP column passes of 4 * GPR channels): the raw values stay in registers as the unit's residual, their SnakeBeta
// goes into the swizzled bf16 tile.  All loads are issued before the first value is used.
template <int RT, int GPR, int NP>
__device__ __forceinline__ void tile_load_snake_keep(uint4* tile, const float* src, int C, int vpr, int row0, const float* sn,
                                                     float (&res)[NP][RT / (256 / GPR)][4]) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int row = row0 + k * RPI + r0, col = p * 4 * GPR + 4 * c4;
            const float4 x = *reinterpret_cast<const float4*>(src + (long)row * C + col);
            res[p][k][0] = x.x; res[p][k][1] = x.y; res[p][k][2] = x.z; res[p][k][3] = x.w;
        }
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int col = p * 4 * GPR + 4 * c4;
        const float4 al = *reinterpret_cast<const float4*>(sn + col), dn = *reinterpret_cast<const float4*>(sn + C + col);
        const float alv[4] = {al.x, al.y, al.z, al.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int row = row0 + k * RPI + r0;
            u16 h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(snake_apply<bf16_t>(res[p][k][q], alv[q], dnv[q]));
            char* dst = reinterpret_cast<char*>(tile + row * vpr + tile_pcol(row, col / 8, vpr)) + (col % 8) * 2;
            *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        }
    }
}

// The same pass in two halves for kernels that have work to put between the residual loads and their use.
template <int U>
struct EpiRows {
    float rr[U][4];
    long off[U];
    int tp[U];
};
template <typename T, int RT, int GPR>
__device__ __forceinline__ void epi_rows_prefetch(const ConvEpi& e, const EpiCols& pc, const int* row_u, const int* row_tp,
                                                  EpiRows<(RT / (256 / GPR) >= 8 ? 8 : RT / (256 / GPR))>& R) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI, U = IT >= 8 ? 8 : IT;
    const int r0 = threadIdx.x / GPR;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        const int row = k * RPI + r0;
        R.tp[k] = row_tp[row];
        R.off[k] = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) R.rr[k][q] = 0.f;
        if (R.tp[k] >= 0) {
            R.off[k] = conv_epi_offset(e, pc, row_u[row], R.tp[k]);
            conv_epi_resid<T>(e, R.off[k], R.rr[k]);
        }
    }
}
template <typename T, int RT, int GPR, int OP>
__device__ __forceinline__ void epi_rows_finish(const ConvEpi& e, const EpiCols& pc, const float* otile, const int* row_u,
                                                EpiRows<(RT / (256 / GPR) >= 8 ? 8 : RT / (256 / GPR))>& R) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI, U = IT >= 8 ? 8 : IT;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
#pragma unroll
    for (int k = 0; k < U; ++k) {
        if (R.tp[k] < 0) continue;
        const int row = k * RPI + r0;
        const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
        float v[4] = {a4.x, a4.y, a4.z, a4.w};
        conv_epi_finish<T>(e, pc, row_u[row], R.off[k], v, R.rr[k]);
    }
}

template <typename T, int MT, int NT, bool STREAM = false>
__global__ __launch_bounds__(256) void conv_halo_kernel(const T* __restrict__ W, int wpk, const T* __restrict__ X, int K, int C, int taps, int dil,
                                                        int xcd, ConvEpi e) {
    using F = CFrag<T>;
    extern __shared__ uint4 tile[];
    constexpr int RT = 16 * MT;
    constexpr int VPE = 16 / (int)sizeof(T);  // elements per 16-byte vector
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = frag_row(lane), g = lane >> 4;  // r: the lane's row of a 16-row fragment
    int colb;
    long rowb;
    if (!conv_block_tile((e.M + RT - 1) / RT * (e.N / (64 * NT)), e.N / (64 * NT), xcd, colb, rowb)) return;
    const long m0 = rowb * RT;
    const int vpr = C / VPE;  // vectors per row (a multiple of 8)
    const int rows_t = RT + (taps - 1) * dil;
    tile_load_dma<T>(tile, X + m0 * C, C, vpr, rows_t * vpr);
    // wave w owns the 16-column tiles 4 j + w of the block's 64 NT columns (interleaved, so that an epilogue pass over
    // tiles j .. j + PT - 1 of every wave covers one contiguous run of columns)
    const int nb = colb * (64 * NT);
    const int cpt = C / F::CHUNK, nC = taps * cpt;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PF = MT * NT >= 12 ? 2 : 4;  // weight chunks in flight per wave (register budget)
    F a[PF][NT];
    WFragPtr<T> wrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wrow[j] = WFragPtr<T>(W, nb + (4 * j + wave) * 16, K, lane, wpk);
    // (weight chunks past the end of the K walk are clamped to the last one, not branched around: with a branch between the loads
    // and their use the compiler cannot count the loads in flight, and every wait becomes a wait for all of them)
#pragma unroll
    for (int s = 0; s < PF - 1; ++s) {
#pragma unroll
        for (int j = 0; j < NT; ++j) a[s][j].load(wrow[j].chunk(min(s, nC - 1)), wrow[j].hs);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of the tile has landed before it signals the barrier
    __syncthreads();
    // The fragment halves ("pieces": 16 bytes per lane, NT MFMAs each) go through a ring of four, read LDS_AHEAD pieces before their
    // MFMAs and across step boundaries; (tap, c) of a step are counted along instead of divided out.
    constexpr int PPS = 2 * MT, LDS_AHEAD = 3;  // pieces per step
    static_assert(PPS % 4 == 0, "ring of four pieces");
    auto row_slot = [&](int tap, int c) {
        const int vb = (c * F::CHUNK + g * F::EPL) / VPE;  // first of the two 16-byte vectors of this lane's fragment
        return (r + tap * dil) * vpr + tile_pcol(r + tap * dil, vb, vpr);  // (16 rows further: the same slot of the row)
    };
    int tap = 0, c = 0, xr = row_slot(0, 0);
    uint4 hq[4];
#pragma unroll
    for (int pp = 0; pp < LDS_AHEAD; ++pp) hq[pp] = tile[(16 * (pp >> 1) * vpr + xr) ^ (pp & 1)];
    for (int q0 = 0; q0 < nC; q0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int q = q0 + s;
            if (q >= nC) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j) a[(s + PF - 1) % PF][j].load(wrow[j].chunk(min(q + PF - 1, nC - 1)), wrow[j].hs);
            __builtin_amdgcn_sched_group_barrier(0x020, 2 * NT, 0);
            int ntap = tap, nc = c + 1;
            if (nc == cpt) { nc = 0; ++ntap; }
            if (q + 1 >= nC) { ntap = tap; nc = c; }  // (the last step reads its own first pieces again)
            const int xn = row_slot(ntap, nc);
#pragma unroll
            for (int pp = 0; pp < PPS; ++pp) {
                const int np = pp + LDS_AHEAD;
                hq[np % 4] = np < PPS ? tile[(16 * (np >> 1) * vpr + xr) ^ (np & 1)] : tile[(16 * ((np - PPS) >> 1) * vpr + xn) ^ (np & 1)];
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[pp >> 1][j] = F::mma_half(a[s][j], pp & 1, hq[pp % 4], acc[pp >> 1][j]);
                __builtin_amdgcn_sched_group_barrier(0x008, NT * F::HALF_MFMAS, 0);
            }
            tap = ntap;
            c = nc;
            xr = xn;
        }
    }
    // Epilogue through LDS: the accumulators (a lane holds 4 columns of one row, 16 rows per wave) are transposed so that
    // consecutive lanes own consecutive columns of the same row and every global access of the tail is a full contiguous run.
    // PT column tiles per wave go through the (re-used) LDS tile per pass, so that the staging buffer stays near 34 KB.
    constexpr int PT = (MT >= 8 || NT == 1) ? 1 : 2;
    constexpr int CP = 64 * PT, OP = CP + 4;  // columns per pass, padded f32 row pitch
    static_assert(NT % PT == 0, "column tiles per wave must be a multiple of the tiles per epilogue pass");
    float* otile = reinterpret_cast<float*>(tile);
    __shared__ long row_mem[RT];  // per tile row: strip and step (step < 0: the row holds no output), or its output offset
    int* row_u = reinterpret_cast<int*>(row_mem);
    int* row_tp = row_u + RT;
    if constexpr (STREAM) {
        epi_row_offsets<RT>(e, m0, row_mem);
    } else if ((int)threadIdx.x < RT) {
        const long m = m0 + threadIdx.x;
        int uu = 0, tt = -1;
        if (m < e.M) {
            const long u = m / e.S_in;
            const int tp = (int)(m - u * e.S_in);
            uu = (int)u;
            tt = tp < e.lens[u] ? tp : -1;
        }
        row_u[threadIdx.x] = uu;
        row_tp[threadIdx.x] = tt;
    }
#pragma unroll
    for (int p = 0; p < NT / PT; ++p) {
        __syncthreads();  // the input tile / the previous pass is no longer read
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < PT; ++jj)
                *reinterpret_cast<f32x4_t*>(otile + (16 * i + r) * OP + (4 * jj + wave) * 16 + 4 * g) = acc[i][p * PT + jj];
        __syncthreads();
        if constexpr (STREAM) epi_pass_stream<T, RT, CP / 4, OP>(e, otile, row_mem, nb + p * CP);
        else epi_pass<T, RT, CP / 4, OP>(e, otile, row_u, row_tp, nb + p * CP);
    }
}

// Second tiling for long inputs: the four waves form a WM x WN grid, each wave owns 16*MT rows x 16*NT columns, so a weight
// fragment feeds MT MFMAs and an input fragment NT MFMAs.  At MT = 8, NT = 4 the LDS reads per MFMA are a quarter and the
// weight loads per MFMA half of those of the kernel above (where all four waves read the same input fragments), which is
// what bounds it.  The input channels are walked in phases of `cph` channels so that the tile of 256 rows + halo stays in LDS
// whatever the layer's width; the weight stream is prefetched across phase boundaries.
template <typename T, int MT, int NT, int WN, bool STREAM = false>
__global__ __launch_bounds__(256, (MT * NT <= 32 && sizeof(T) == 2) ? 2 : 1) void conv_tile_kernel(const T* __restrict__ W, int wpk, const T* __restrict__ X, int K, int C, int taps, int dil,
                                                        int cph, long x_lo, int xcd, ConvEpi e) {
    using F = CFrag<T>;
    extern __shared__ uint4 tile[];
    constexpr int WM = 4 / WN;
    constexpr int RT = 16 * MT * WM;
    constexpr int VPE = 16 / (int)sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = frag_row(lane), g = lane >> 4;  // r: the lane's row of a 16-row fragment
    const int wm = wave / WN, wn = wave - wm * WN;
    int colb;
    long rowb;
    if (!conv_block_tile((e.M + RT - 1) / RT * (e.N / (16 * NT * WN)), e.N / (16 * NT * WN), xcd, colb, rowb)) return;
    const long m0 = rowb * RT;
    const int vpr = cph / VPE;  // vectors per tile row (a multiple of 8)
    const int rows_t = RT + (taps - 1) * dil;
    const int nb = colb * (16 * NT * WN);
    // x_lo != 0: X carries a remainder plane x_lo elements behind it (store_split); its phases follow the value plane's and
    // meet the same weights, so the product sees the activations to about 16 significant bits
    const int cpp = cph / F::CHUNK, per_phase = taps * cpp, nPh = C / cph, nC = (x_lo ? 2 : 1) * nPh * per_phase;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PF = 2;  // weight chunks in flight per wave
    F a[PF][NT];
    WFragPtr<T> wrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wrow[j] = WFragPtr<T>(W, nb + (j * WN + wn) * 16, K, lane, wpk);
    // the chunk of step (phase ph, tap, chunk c inside the phase's channels) in a weight row: (tap * C + (ph % nPh) * cph) / CHUNK + c
#pragma unroll
    for (int j = 0; j < NT; ++j) a[0][j].load(wrow[j].chunk(0), wrow[j].hs);
    // Phases outside, their steps inside: the tile load's loop (a number of loads the compiler cannot count) stays out of the
    // loop that streams the weights, where the wait before a step's MFMAs then leaves the chunk requested for the next step in
    // flight (with both in one loop every other step waited for the loads it had just issued).
    const int nPhT = nC / per_phase;
    QTTS_TS(0);
    for (int ph = 0; ph < nPhT; ++ph) {
        // the phase's channels of the input tile (the first chunk of its weights is already on its way)
        __syncthreads();
        tile_load_dma<T>(tile, X + (ph >= nPh ? x_lo : 0L) + m0 * C + (size_t)(ph % nPh) * cph, C, vpr, rows_t * vpr);
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (ph == 0) QTTS_TS(1);
        if (ph == 1) QTTS_TS(2);
        // (tap, c) of the step and the weight chunk of the next one are counted along: divisions here would be a chain of some
        // seventy scalar instructions per step in front of the wave's MFMAs
        const int base = (ph % nPh) * (cph / F::CHUNK), base_next = ((ph + 1 < nPhT ? ph + 1 : 0) % nPh) * (cph / F::CHUNK);
        int tap = 0, c = 0;
        for (int rem0 = 0; rem0 < per_phase; rem0 += PF) {
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                const int rem = rem0 + s;
                if (rem >= per_phase) continue;
                int ntap = tap, nc = c + 1;
                if (nc == cpp) { nc = 0; ++ntap; }
                {   // the next step's weights (the last step of all requests the first chunk again: no branch around the loads)
                    const int o = rem + 1 < per_phase ? ntap * (C / F::CHUNK) + base + nc : base_next;
#pragma unroll
                    for (int j = 0; j < NT; ++j) a[(s + 1) % PF][j].load(wrow[j].chunk(o), wrow[j].hs);
                }
                __builtin_amdgcn_sched_group_barrier(0x020, 2 * NT, 0);  // (first in the step: they have the whole step to arrive)
                const int vb = (c * F::CHUNK + g * F::EPL) / VPE;
                const int xr = (wm * (16 * MT) + r + tap * dil) * vpr + tile_pcol(r + tap * dil, vb, vpr);  // (16 rows further: the same slot of the row)
                // The 2 * MT fragment halves of the step are read LDS_AHEAD halves before their MFMAs: one half ahead (what the
                // compiler schedules by itself) leaves every read's latency to the other wave of the SIMD to cover.
                constexpr int NH = 2 * MT, LDS_AHEAD = 3;
                uint4 hq[NH];
#pragma unroll
                for (int t = 0; t < LDS_AHEAD; ++t) hq[t] = tile[(16 * (t >> 1) * vpr + xr) ^ (t & 1)];
                __builtin_amdgcn_sched_group_barrier(0x100, LDS_AHEAD, 0);
#pragma unroll
                for (int t = 0; t < NH; ++t) {
                    if (t + LDS_AHEAD < NH) {
                        hq[t + LDS_AHEAD] = tile[(16 * ((t + LDS_AHEAD) >> 1) * vpr + xr) ^ ((t + LDS_AHEAD) & 1)];
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[t >> 1][j] = F::mma_half(a[s][j], t & 1, hq[t], acc[t >> 1][j]);
                    __builtin_amdgcn_sched_group_barrier(0x008, NT * F::HALF_MFMAS, 0);
                }
                tap = ntap;
                c = nc;
            }
        }
        if (per_phase % PF) {  // an odd number of steps: the chunk in flight sits in the other half of the ring
#pragma unroll
            for (int j = 0; j < NT; ++j) a[0][j] = a[1][j];
        }
    }
    // epilogue through LDS, PT column tiles of every wave per pass (see conv_halo_kernel)
    QTTS_TS(3);
    constexpr int PT = NT % 2 == 0 ? 2 : 1;
    constexpr int CP = 16 * WN * PT, OP = CP + 4;
    float* otile = reinterpret_cast<float*>(tile);
    __shared__ long row_mem[RT];  // per tile row: strip and step, or (STREAM) its output offset
    int* row_u = reinterpret_cast<int*>(row_mem);
    int* row_tp = row_u + RT;
    if constexpr (STREAM) {
        epi_row_offsets<RT>(e, m0, row_mem);
    } else {
        for (int t = threadIdx.x; t < RT; t += 256) {
            const long m = m0 + t;
            int uu = 0, tt = -1;
            if (m < e.M) {
                const long u = m / e.S_in;
                const int tp = (int)(m - u * e.S_in);
                uu = (int)u;
                tt = tp < e.lens[u] ? tp : -1;
            }
            row_u[t] = uu;
            row_tp[t] = tt;
        }
    }
#pragma unroll
    for (int p = 0; p < NT / PT; ++p) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < PT; ++jj)
                *reinterpret_cast<f32x4_t*>(otile + (wm * (16 * MT) + 16 * i + r) * OP + (jj * WN + wn) * 16 + 4 * g) = acc[i][p * PT + jj];
        __syncthreads();
        if constexpr (STREAM) epi_pass_stream<T, RT, CP / 4, OP>(e, otile, row_mem, nb + p * CP);
        else epi_pass<T, RT, CP / 4, OP>(e, otile, row_u, row_tp, nb + p * CP);
        if (p == 0) QTTS_TS(4);
    }
    QTTS_TS(5);
}

template <typename T, int MT, int NT, int WN>
void launch_tile(hipStream_t st, const T* W, int wpk, const T* X, int K, int C, int taps, int dil, int cph, long x_lo, const ConvEpi& e) {
    constexpr int RT = 16 * MT * (4 / WN);
    constexpr int PT = NT % 2 == 0 ? 2 : 1;
    const size_t lds = (std::max((size_t)(RT + (taps - 1) * dil) * cph * sizeof(T), (size_t)RT * (16 * WN * PT + 4) * 4) + 1023) / 1024 * 1024;
    const long gy = (e.M + RT - 1) / RT;
    const int xcd = conv_xcd_order();
    unsigned grid = 0;
    QTTS_CHECK(conv_grid(gy * (e.N / (16 * NT * WN)), xcd, grid), ERR_SHAPE, "conv: too many tiles for one launch");
    auto* fn = conv_epi_stream_ok(e) ? conv_tile_kernel<T, MT, NT, WN, true> : conv_tile_kernel<T, MT, NT, WN, false>;
    // the same ceiling from every caller: streams of several host threads launch this kernel with different tile sizes
    QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CEILING));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, W, wpk, X, K, C, taps, dil, cph, x_lo, xcd, e);
}

// 256 x 128 or 256 x 192 tiles.  The choice depends on the layer only, never on the number of rows: a row's result must not
// depend on what else is in the batch (the two kernels walk K in different orders).
template <typename T>
bool try_tile(hipStream_t st, const T* W, int wpk, const T* X, int K, int C, int taps, int dil, const ConvEpi& e, long x_lo = 0) {
    static const bool off = getenv("QTTS_CONV_TILE2") && atoi(getenv("QTTS_CONV_TILE2")) == 0;
    static const long min_rows = getenv("QTTS_CONV_TILE2_ROWS") ? atol(getenv("QTTS_CONV_TILE2_ROWS")) : 1;
    static const long min_k = getenv("QTTS_CONV_TILE2_K") ? atol(getenv("QTTS_CONV_TILE2_K")) : 2600;
    const long halo = (long)(taps - 1) * dil;
    // at most two blocks per CU (221 registers per lane, up to 80 KB of LDS): little hides its tile load and its epilogue, so it
    // only wins where the K walk is long
    // (measured on the vocoder: wins for 7-tap convs over >= 384 channels, for the wide two-tap transposed convs, and where the
    // halo of a dilated conv would push the other kernel's tile past 64 KB; no gain on anything narrower or on 1x1 convs)
    const bool long_k = (long)taps * C * (x_lo ? 2 : 1) >= min_k;  // (a remainder plane doubles the K walk)
    const bool wide_ct = taps == 2 && (long)taps * C >= 1536 && e.N >= 1920;
    const bool big_halo = (size_t)(64 + halo) * C * sizeof(T) > (size_t)64 * 1024;
    if (off || e.M < min_rows || halo > 128 || 256 + halo + 8 > CONV_HALO_SLACK || (e.M + 255) / 256 > 0x7fffffffL ||
        !(long_k || wide_ct || big_halo))
        return false;
    // channels per phase: as many as keep the tile of 256 rows + halo within 80 KB (bf16: two blocks per CU, so that one
    // block's tile load and epilogue run under the other's K walk) or 96 KB (f32: the accumulators leave room for one block)
    const size_t tile_cap = (size_t)(sizeof(T) == 2 ? 80 : 96) * 1024;
    int cph = 0;
    for (int c = C; c >= 64; c -= 64)
        if (C % c == 0 && (size_t)(256 + halo) * c * sizeof(T) <= tile_cap) { cph = c; break; }
    if (cph == 0) return false;
    if (e.N % 128 == 0) launch_tile<T, 8, 4, 2>(st, W, wpk, X, K, C, taps, dil, cph, x_lo, e);
    else if (e.N % 192 == 0) launch_tile<T, 8, 6, 2>(st, W, wpk, X, K, C, taps, dil, cph, x_lo, e);
    else return false;
    return true;
}

// One residual unit of the vocoder in one launch (audio/decoder/residual_unit.rs:11-45): x + conv1x1(Snake(conv7(xa))), where
// xa = Snake(x) is the activated copy the previous layer left.  The 7-tap conv runs as in conv_halo_kernel; its tail
// (bias, SnakeBeta) goes from the accumulators straight into a second LDS tile laid out as an MFMA B operand, the 1x1 conv is a
// second, short K walk over that tile, and only then comes the epilogue that touches memory (f32 residual in / out, activated
// copy for the next layer).  The intermediate tensor never leaves the CU, and the HBM-bound tail of one block overlaps the
// MFMA phases of the others.  A block owns all N = C output channels (NT = C / 64 column tiles per wave).
// column tiles (of 64) per epilogue pass of the fused unit: two where the f32 staging tile then still fits beside the others
constexpr int unit_pt(int MT, int NT) { return (NT % 2 == 0 && MT <= 4) ? 2 : 1; }

// F32IN (bf16 only): X is null and the input tile is made from the f32 stream Xf with the input activation (alpha1, den1).
template <typename T, int MT, int NT, bool F32IN = false>
__global__ __launch_bounds__(256, (NT == 3 && sizeof(T) == 2) ? 2 : 1) void conv_unit_kernel(const T* __restrict__ W1, const T* __restrict__ W2, int wpk, const T* __restrict__ X, int taps,
                                                        int dil, const float* __restrict__ bias1, const float* __restrict__ alpha2,
                                                        const float* __restrict__ den2, int xcd, ConvEpi e, const float* __restrict__ Xf,
                                                        const float* __restrict__ alpha1, const float* __restrict__ den1) {
    using F = CFrag<T>;
    extern __shared__ uint4 tile[];
    constexpr int RT = 16 * MT;
    constexpr int C = 64 * NT;
    constexpr int VPE = 16 / (int)sizeof(T);
    constexpr int vpr = C / VPE;
    constexpr int cpt = C / F::CHUNK;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = frag_row(lane), g = lane >> 4;  // r: the lane's row of a 16-row fragment
    int colb;
    long rowb;
    if (!conv_block_tile((e.M + RT - 1) / RT, 1, xcd, colb, rowb)) return;
    const long m0 = rowb * RT;
    const int rows_t = RT + (taps - 1) * dil;
    const int K = taps * C, nC = taps * cpt;
    const int tile_slots = (max(rows_t * vpr * 16, RT * (64 * unit_pt(MT, NT) + 4) * 4) + 1023) / 1024 * 64;
    uint4* htile = tile + tile_slots;
    if constexpr (!F32IN) tile_load_dma<T>(tile, X + m0 * C, C, vpr, rows_t * vpr);
    constexpr int PT = unit_pt(MT, NT);
    constexpr int CP = 64 * PT, OP = CP + 4, GPR = CP / 4;
    constexpr int EU = (RT / (256 / GPR) >= 8 ? 8 : RT / (256 / GPR));
    // prefetch the first pass's residual rows where that costs no occupancy (at NT = 2 it would take the third block off a CU)
    constexpr bool ONE_ROUND = RT / (256 / GPR) == EU && sizeof(T) == 2 && NT >= 3 && !F32IN;
    // F32IN: the residual rows of the tile, read once when the input tile is built (epilogue thread layout)
    float res[F32IN ? NT / PT : 1][RT / (256 / GPR)][4];
    __shared__ int row_u[RT], row_tp[RT];
    if ((int)threadIdx.x < RT) {
        const long m = m0 + threadIdx.x;
        int uu = 0, tt = -1;
        if (m < e.M) {
            const long u = m / e.S_in;
            const int tp = (int)(m - u * e.S_in);
            uu = (int)u;
            tt = tp < e.lens[u] ? tp : -1;
        }
        row_u[threadIdx.x] = uu;
        row_tp[threadIdx.x] = tt;
    }
    if constexpr (F32IN) {
        __shared__ float sn1[2 * C];
        for (int c = threadIdx.x; c < C; c += 256) {
            float al = alpha1[c], dn = den1[c];
            snake_prep<bf16_t>(al, dn);
            sn1[c] = al;
            sn1[C + c] = dn;
        }
        __syncthreads();
        // halo rows as 16-byte slots; the tile's own rows in the epilogue's layout, their raw values kept as the residual
        const int halo = (taps - 1) * dil;
        tile_load_snake(tile, Xf + m0 * C, C, vpr, halo * vpr, sn1);
        tile_load_snake_keep<RT, GPR, NT / PT>(tile, Xf + m0 * C, C, vpr, halo, sn1, res);
    }
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    constexpr int PF = (MT * NT >= 12 || F32IN) ? 2 : 4;  // (F32IN: the residual rows take the registers of two stages)
    F a[PF][NT];
    WFragPtr<T> wrow[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wrow[j] = WFragPtr<T>(W1, (4 * j + wave) * 16, K, lane, wpk);
#pragma unroll
    for (int s = 0; s < PF - 1; ++s)
        if (s < nC) {
#pragma unroll
            for (int j = 0; j < NT; ++j) a[s][j].load(wrow[j].chunk(s), wrow[j].hs);
        }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
    __syncthreads();
    for (int q0 = 0; q0 < nC; q0 += PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            const int q = q0 + s;
            if (q + PF - 1 < nC) {
#pragma unroll
                for (int j = 0; j < NT; ++j) a[(s + PF - 1) % PF][j].load(wrow[j].chunk(q + PF - 1), wrow[j].hs);
            }
            if (q < nC) {
                const int tap = q / cpt, c = q - tap * cpt;
                const int vb = (c * F::CHUNK + g * F::EPL) / VPE;
                const int xr = (r + tap * dil) * vpr + tile_pcol(r + tap * dil, vb, vpr);  // (16 rows further: the same slot of the row)
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const int x0 = 16 * i * vpr + xr;
                    const uint4 h0 = tile[x0], h1 = tile[x0 ^ 1];
                    F b;
                    __builtin_memcpy(&b, &h0, 16);
                    __builtin_memcpy((char*)&b + 16, &h1, 16);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[s][j], b, acc[i][j]);
                }
            }
        }
    }
    // residual rows of the first epilogue pass and the weights of the 1x1 conv: in flight while the tail of the 7-tap conv is computed
    EpiCols pc0;
    EpiRows<EU> R0;
    if constexpr (ONE_ROUND) {
        conv_epi_cols<T>(e, 4 * ((int)threadIdx.x % GPR), pc0);
        epi_rows_prefetch<T, RT, GPR>(e, pc0, row_u, row_tp, R0);
    }
    F a2[cpt][NT];
#pragma unroll
    for (int c = 0; c < cpt; ++c)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const WFragPtr<T> w2(W2, (4 * j + wave) * 16, C, lane, wpk);
            a2[c][j].load(w2.chunk(c), w2.hs);
        }
    // tail of the 7-tap conv: bias, SnakeBeta of the value the unfused layer would have stored, into the second tile.  A lane
    // holds four consecutive channels of one row: one 8- or 16-byte store into the swizzled row-major layout.
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = (4 * j + wave) * 16 + 4 * g;
        float bs[4], al[4], dn[4];
        ld4(bias1 + n, bs);
        ld4(alpha2 + n, al);
        ld4(den2 + n, dn);
#pragma unroll
        for (int q = 0; q < 4; ++q) snake_prep<T>(al[q], dn[q]);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int row = 16 * i + r;
            T y[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float x = to_f32<T>(from_f32<T>(acc[i][j][q] + bs[q]));
                y[q] = from_f32<T>(snake_apply<T>(x, al[q], dn[q]));
            }
            char* dst = reinterpret_cast<char*>(htile + row * vpr + tile_pcol(row, n / VPE, vpr)) + (n % VPE) * sizeof(T);
            __builtin_memcpy(dst, y, 4 * sizeof(T));
            acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < cpt; ++c) {
        const int vb = (c * F::CHUNK + g * F::EPL) / VPE;
        const int xr = r * vpr + tile_pcol(r, vb, vpr);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int x0 = 16 * i * vpr + xr;
            const uint4 h0 = htile[x0], h1 = htile[x0 ^ 1];
            F b;
            __builtin_memcpy(&b, &h0, 16);
            __builtin_memcpy((char*)&b + 16, &h1, 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a2[c][j], b, acc[i][j]);
        }
    }
    // epilogue of the 1x1 conv through LDS (the input tile is dead: every wave passed the barrier above after its last read)
    float* otile = reinterpret_cast<float*>(tile);
#pragma unroll
    for (int p = 0; p < NT / PT; ++p) {
        if (p > 0) __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int jj = 0; jj < PT; ++jj)
                *reinterpret_cast<f32x4_t*>(otile + (16 * i + r) * OP + (4 * jj + wave) * 16 + 4 * g) = acc[i][p * PT + jj];
        __syncthreads();
        if constexpr (F32IN) epi_pass_regs<T, RT, GPR, OP>(e, otile, row_u, row_tp, p * CP, res[p]);
        else if (ONE_ROUND && p == 0) epi_rows_finish<T, RT, GPR, OP>(e, pc0, otile, row_u, R0);
        else epi_pass<T, RT, GPR, OP>(e, otile, row_u, row_tp, p * CP);
    }
}

// The fused unit as the vocoder's wide blocks run it - bf16 matrix cores, seven taps, the f32 stream in and out (F32IN above) -
// as a persistent kernel: a workgroup walks a run of consecutive 64-row tiles.
//  * The dilation is a template parameter and the K walk is unrolled over its 7 * NT chunks: the LDS addresses of the input rows
//    are immediates behind one per-lane base, the loop carries no address arithmetic.
//  * The causal halo of a tile (its first 6 * DIL rows) is the tail of the previous tile's rows, already activated and in LDS:
//    it is moved down inside LDS, and only the tile's own 64 rows are read from memory - in the epilogue's thread layout, so the
//    raw values stay in registers as the unit's residual (the stream is read exactly once).
//  * PREFETCH: the next tile's rows are requested before the current tile's K walk and arrive under it.
//  * The epilogue is the unit's own (bias, residual, f32 store and / or SnakeBeta of the next layer as one 8-byte store per plane).
// Instruction issue and exposed memory latency, not the matrix pipe or bandwidth, bound the one-tile-per-workgroup kernel at
// these widths (about 1900 vector instructions per wave and tile beside 256 matrix instructions, waves parked at s_waitcnt for
// 46 % of their cycles).
// Needs e.r == 1, S_in == S_out, ld_out == C, no activation / scale / per-strip bias, resid_f32 = the input stream.
template <int RT, int GPR, int NP>
__device__ __forceinline__ void unit_rows_load(const float* src, int C, float (&res)[NP][RT / (256 / GPR)][4]) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const float4 x = *reinterpret_cast<const float4*>(src + (long)(k * RPI + r0) * C + p * 4 * GPR + 4 * c4);
            res[p][k][0] = x.x; res[p][k][1] = x.y; res[p][k][2] = x.z; res[p][k][3] = x.w;
        }
}
template <int RT, int GPR, int NP>
__device__ __forceinline__ void unit_rows_store(uint4* tile, int C, int vpr, int row0, const float* sn, const float (&res)[NP][RT / (256 / GPR)][4]) {
    constexpr int RPI = 256 / GPR, IT = RT / RPI;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const int col = p * 4 * GPR + 4 * c4;
        const float4 al = *reinterpret_cast<const float4*>(sn + col), dn = *reinterpret_cast<const float4*>(sn + C + col);
        const float alv[4] = {al.x, al.y, al.z, al.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int row = row0 + k * RPI + r0;
            u16 h[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(snake_apply<bf16_t>(res[p][k][q], alv[q], dnv[q]));
            char* dst = reinterpret_cast<char*>(tile + row * vpr + tile_pcol(row, col / 8, vpr)) + (col % 8) * 2;
            *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
        }
    }
}

template <int NT, int DIL, int PF, int PT, int BPC, bool PREFETCH>
__global__ __launch_bounds__(256, BPC) void conv_unit_f32_kernel(const bf16_t* __restrict__ W1, const bf16_t* __restrict__ W2, int wpk,
                                                                 const float* __restrict__ bias1, const float* __restrict__ alpha2,
                                                                 const float* __restrict__ den2, int tiles_per_block, ConvEpi e,
                                                                 const float* __restrict__ Xf, const float* __restrict__ alpha1,
                                                                 const float* __restrict__ den1) {
    using T = bf16_t;
    using F = CFrag<T>;
    extern __shared__ uint4 tile[];
    constexpr int MT = 4, RT = 64, C = 64 * NT, TAPS = 7, HALO = (TAPS - 1) * DIL, ROWS_T = RT + HALO;
    constexpr int vpr = C / 8, cpt = NT, nC = TAPS * cpt, K = TAPS * C;
    constexpr int CP = 64 * PT, OP = CP + 4, GPR = CP / 4, NP = NT / PT, RPI = 256 / GPR, IT = RT / RPI;
    static_assert(NT % PT == 0 && HALO < RT, "unit geometry");
    constexpr int tile_slots = (ROWS_T * vpr * 16 + 1023) / 1024 * 64;  // input tile; behind it the second tile / the f32 staging tile
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = frag_row(lane), g = lane >> 4;  // r: the lane's row of a 16-row fragment
    const long total = (e.M + RT - 1) / RT;
    const long t0 = (long)blockIdx.x * tiles_per_block, t1 = t0 + tiles_per_block < total ? t0 + tiles_per_block : total;
    if (t0 >= total) return;
    uint4* htile = tile + tile_slots;
    float* otile = reinterpret_cast<float*>(htile);
    __shared__ int row_ok[2][RT];
    __shared__ float sn1[2 * C];
    // per-channel parameters of the tail and of the epilogue as snake_prep leaves them: read from memory once per workgroup instead
    // of once per tile, where each read was a round trip in front of the tail and of the epilogue
    __shared__ float prm[6 * C];
    for (int c = threadIdx.x; c < C; c += 256) {
        float al = alpha1[c], dn = den1[c];
        snake_prep<T>(al, dn);
        sn1[c] = al;
        sn1[C + c] = dn;
        prm[c] = bias1[c];
        al = alpha2[c]; dn = den2[c];
        snake_prep<T>(al, dn);
        prm[C + c] = al;
        prm[2 * C + c] = dn;
        prm[3 * C + c] = e.bias[c];
        al = 0.f; dn = 0.f;
        if (e.out2) { al = e.alpha_exp[c]; dn = e.beta_den[c]; snake_prep<T>(al, dn); }
        prm[4 * C + c] = al;
        prm[5 * C + c] = dn;
    }
    float res[NP][IT][4], resn[PREFETCH ? NP : 1][PREFETCH ? IT : 1][4];
    unit_rows_load<RT, GPR, NP>(Xf + (t0 * RT + HALO) * C, C, res);
    __syncthreads();
    tile_load_snake(tile, Xf + t0 * RT * C, C, vpr, HALO * vpr, sn1);  // the first tile's halo comes from memory
    const uint4* lrow = tile + r * vpr;  // the lane's row of every 16-row tile; taps and row tiles are constant offsets from it
    const uint4* hrow = htile + r * vpr;
    const int c4 = threadIdx.x % GPR, r0 = threadIdx.x / GPR;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    for (long t = t0; t < t1; ++t) {
        const long m0 = t * RT;
        // Per-channel parameters and weight addresses are the same for every tile; hoisted out of this loop they would sit in
        // registers across the K walk and push the kernel into scratch.  Passing the pointers through an empty asm statement
        // makes them opaque per iteration: every tile re-derives its addresses (a handful of scalar instructions) and re-reads
        // the parameters from cache where it needs them.
        // (an opaque zero offset rather than opaque pointers: a pointer that went through the asm statement is no longer known
        // to point to global memory, and its loads become flat loads, which count on both the vector-memory and the LDS counter
        // and force every wait around them to drain both)
        long z = 0;
        asm volatile("" : "+s"(z));
        const bf16_t *W1l = W1 + z, *W2l = W2 + z;
        WFragPtr<T> wrow[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) wrow[j] = WFragPtr<T>(W1l, (4 * j + wave_s) * 16, K, lane, wpk);
        int* ok = row_ok[t & 1];
        if ((int)threadIdx.x < RT) {
            const long m = m0 + threadIdx.x;
            int v = 0;
            if (m < e.M) {
                const long u = m / e.S_in;
                v = (int)(m - u * e.S_in) < e.lens[u];
            }
            ok[threadIdx.x] = v;
        }
        if constexpr (!PREFETCH) {
            if (t > t0) unit_rows_load<RT, GPR, NP>(Xf + (m0 + HALO) * C, C, res);
        }
        F a[PF][NT];
#pragma unroll
        for (int s = 0; s < PF - 1; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j) a[s][j].load(wrow[j].chunk(s), wrow[j].hs);
        unit_rows_store<RT, GPR, NP>(tile, C, vpr, HALO, sn1, res);
        if constexpr (PREFETCH) {
            if (t + 1 < t1) unit_rows_load<RT, GPR, NP>(Xf + (m0 + RT + HALO) * C, C, resn);
        }
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        // The walk in pieces of one 16-byte fragment half (NT MFMAs), each a scheduling region of its own: a step's first piece
        // issues the weight loads for PF - 1 steps ahead, every piece the LDS read of the piece LDS_AHEAD further on, then its
        // MFMAs.  (In one region the scheduler sinks loads and reads next to their uses, and every MFMA group starts with a full
        // LDS round trip.)
        {
            constexpr int PPS = 2 * MT, NPC = nC * PPS, LDS_AHEAD = 3;  // pieces per step, pieces
            auto piece = [&](int p) {
                const int q = p / PPS, i = (p % PPS) >> 1, hh = p & 1;
                const int tap = q / cpt, c = q % cpt;
                const int x0 = tile_pcol(r + tap * DIL, c * 8 + g * 2, vpr);  // (the same for every 16-row fragment of the tile)
                return lrow[(16 * i + tap * DIL) * vpr + (x0 ^ hh)];
            };
            static_assert(PPS % 4 == 0 && LDS_AHEAD < 4, "ring of four pieces");
            uint4 hq[4];  // ring: piece p sits in hq[p % 4]
#pragma unroll
            for (int p = 0; p < LDS_AHEAD; ++p) hq[p] = piece(p);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < nC; ++q) {
#pragma unroll
                for (int pp = 0; pp < PPS; ++pp) {
                    const int p = q * PPS + pp, i = pp >> 1, hh = pp & 1;
                    if (pp == 0 && q + PF - 1 < nC) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) a[(q + PF - 1) % PF][j].load(wrow[j].chunk(q + PF - 1), wrow[j].hs);
                    }
                    if (p + LDS_AHEAD < NPC) hq[(pp + LDS_AHEAD) % 4] = piece(p + LDS_AHEAD);
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = F::mma_half(a[q % PF][j], hh, hq[pp % 4], acc[i][j]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        // weights of the 1x1 conv: in flight while the tail of the 7-tap conv is computed
        constexpr int A2 = cpt < 2 ? cpt : 2;  // (two chunks of them at a time: all NT * NT fragments would not fit beside the rest)
        F a2[A2][NT];
        WFragPtr<T> w2row[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) w2row[j] = WFragPtr<T>(W2l, (4 * j + wave_s) * 16, C, lane, wpk);
#pragma unroll
        for (int c = 0; c < A2; ++c)
#pragma unroll
            for (int j = 0; j < NT; ++j) a2[c][j].load(w2row[j].chunk(c), w2row[j].hs);
        // tail of the 7-tap conv: bias, SnakeBeta of the bf16 value the unfused layer would have stored, into the second tile
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (4 * j + wave) * 16 + 4 * g;
            float bs[4], al[4], dn[4];
            ld4(prm + n, bs);
            ld4(prm + C + n, al);
            ld4(prm + 2 * C + n, dn);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = 16 * i + r;
                u16 h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = bf16_bits_to_f32(f32_to_bf16_bits(acc[i][j][q] + bs[q]));
                    h[q] = f32_to_bf16_bits(snake_apply<T>(x, al[q], dn[q]));
                }
                char* dst = reinterpret_cast<char*>(htile + row * vpr + tile_pcol(row, n / 8, vpr)) + (n % 8) * 2;
                *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();  // second tile complete; every wave is past its last read of the input tile
#pragma unroll
        for (int c = 0; c < cpt; ++c) {
            const int x0 = tile_pcol(r, c * 8 + g * 2, vpr);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const uint4* rp = hrow + 16 * i * vpr;
                const uint4 h0 = rp[x0], h1 = rp[x0 ^ 1];
                F b;
                __builtin_memcpy(&b, &h0, 16);
                __builtin_memcpy((char*)&b + 16, &h1, 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a2[c % A2][j], b, acc[i][j]);
            }
            if (c + A2 < cpt) {
#pragma unroll
                for (int j = 0; j < NT; ++j) a2[c % A2][j].load(w2row[j].chunk(c + A2), w2row[j].hs);
            }
        }
        // the next tile's halo: rows [RT, RT + HALO) move to the front (a shift by a multiple of 64 rows keeps the swizzle)
        if (t + 1 < t1) {
            for (int sidx = threadIdx.x; sidx < HALO * vpr; sidx += 256) tile[sidx] = tile[RT * vpr + sidx];
        }
        __syncthreads();  // every wave is past its reads of the second tile: the f32 staging tile may take its place
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if (p > 0) __syncthreads();
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int jj = 0; jj < PT; ++jj)
                    *reinterpret_cast<f32x4_t*>(otile + (16 * i + r) * OP + (4 * jj + wave) * 16 + 4 * g) = acc[i][p * PT + jj];
            __syncthreads();
            const int col = p * CP + 4 * c4;
            if (e.Cvalid && col >= e.Cvalid) continue;  // columns past the layer's real width are padding (the barriers are block-uniform)
            float b2[4], al[4], dn[4];
            ld4(prm + 3 * C + col, b2);
            ld4(prm + 4 * C + col, al);
            ld4(prm + 5 * C + col, dn);
            const long obase = (m0 + r0 + e.P_out) * C + col;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int row = k * RPI + r0;
                if (!ok[row]) continue;
                const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + 4 * c4);
                const float v[4] = {a4.x + b2[0] + res[p][k][0], a4.y + b2[1] + res[p][k][1], a4.z + b2[2] + res[p][k][2], a4.w + b2[3] + res[p][k][3]};
                const long off = obase + (long)k * RPI * C;
                if (e.out_f32) *reinterpret_cast<float4*>(e.out_f32 + off) = make_float4(v[0], v[1], v[2], v[3]);
                if (e.out2) {
                    float y[4];
                    u16 h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = snake_apply<T>(v[q], al[q], dn[q]);
                        h[q] = f32_to_bf16_bits(y[q]);
                    }
                    T* o = (T*)e.out2 + off;
                    *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                    if (e.out2_lo) {
                        u16 l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) l[q] = f32_to_bf16_bits(y[q] - bf16_bits_to_f32(h[q]));
                        *reinterpret_cast<uint2*>(o + e.out2_lo) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    }
                }
            }
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int p = 0; p < NP; ++p)
#pragma unroll
                for (int k = 0; k < IT; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) res[p][k][q] = resn[p][k][q];
        }
    }
}

// whether a unit's epilogue is the plain one conv_unit_f32_kernel implements
static bool unit_f32_epilogue(const ConvEpi& e, int C) {
    return e.r == 1 && e.S_in == e.S_out && e.ld_out == C && e.Cout == C && e.bias && !e.bias_u && !e.scale && e.act == 0 && !e.resid && e.resid_f32 &&
           !e.out && !e.out_lo && (e.out_f32 || e.out2) && (!e.out2 || (e.act2 != 1 && e.alpha_exp && e.beta_den)) && e.out2_lo % 4 == 0;
}

template <int NT, int DIL, int PF, int PT, int BPC, bool PREFETCH>
static bool launch_unit_f32_cfg(hipStream_t st, const bf16_t* W1, const bf16_t* W2, int wpk, const float* bias1, const float* alpha2, const float* den2,
                                const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1) {
    constexpr int RT = 64, C = 64 * NT, HALO = 6 * DIL;
    constexpr size_t in_b = ((size_t)(RT + HALO) * C * 2 + 1023) / 1024 * 1024;
    constexpr size_t second = (size_t)RT * C * 2, staging = (size_t)RT * (64 * PT + 4) * 4;
    constexpr size_t lds = in_b + (second > staging ? second : staging);
    static_assert(lds * BPC <= (size_t)156 * 1024, "unit tiles of the resident workgroups exceed the LDS of a CU");
    const long total = (e.M + RT - 1) / RT;
    if ((long)RT + HALO + 8 > CONV_HALO_SLACK || total > 0x7fffffffL) return false;
    const long blocks = std::min<long>(total, (long)BPC * 256);
    const int tpb = (int)((total + blocks - 1) / blocks);
    const unsigned grid = (unsigned)((total + tpb - 1) / tpb);
    auto* fn = conv_unit_f32_kernel<NT, DIL, PF, PT, BPC, PREFETCH>;
    QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CEILING));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, W1, W2, wpk, bias1, alpha2, den2, tpb, e, Xf, alpha1, den1);
    return true;
}

// Measured on the vocoder of the benchmark (27 M rows per launch at 128 channels, 9 M at 192; ms per unit, dilations 1 / 3 / 9):
// 128 channels - general kernel 11.5 / 11.8 / 12.5 (10.5 / 11.2 / 11.7 with the residual kept in registers), this kernel with two
// workgroups per CU and prefetch 10.4 / 10.9 / 10.7; 192 channels - general 8.4 / 8.3 / 9.1, this kernel (no prefetch: its
// registers are taken) 8.3 / 7.9 / 7.8.  Skipping one phase at a time shows the phases of a workgroup adding up rather than
// overlapping (18.7 M rows: 7.3 ms whole, K walk 3.2, tail + 1x1 conv 1.0, row loads 0.8, stores 0.8, the rest 1.4), so what
// remains is occupancy: LDS and registers hold two such workgroups per CU.
template <int NT, int DIL>
static bool launch_unit_f32(hipStream_t st, const bf16_t* W1, const bf16_t* W2, int wpk, const float* bias1, const float* alpha2, const float* den2,
                            const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1) {
    if constexpr (NT == 2) return launch_unit_f32_cfg<NT, DIL, 3, 2, 2, true>(st, W1, W2, wpk, bias1, alpha2, den2, e, Xf, alpha1, den1);
    else return launch_unit_f32_cfg<NT, DIL, 2, 1, 2, false>(st, W1, W2, wpk, bias1, alpha2, den2, e, Xf, alpha1, den1);
}

bool conv_unit_supported_fwd(int dtype, int C, int taps, int dil);

struct UnitIn {  // f32 input stream of a fused unit instead of an activated copy (bf16 runs): stream, input activation parameters
    const float* Xf = nullptr;
    const float* alpha1 = nullptr;
    const float* den1 = nullptr;
};

template <typename T, int NT, int MT = 4>
bool launch_unit(hipStream_t st, const T* W1, const T* W2, int wpk, const T* X, int taps, int dil, const float* bias1, const float* alpha2,
                 const float* den2, const ConvEpi& e, const UnitIn& in = UnitIn()) {
    constexpr int RT = 16 * MT, C = 64 * NT, PT = unit_pt(MT, NT);
    const size_t tile_b = (std::max((size_t)(RT + (taps - 1) * dil) * C * sizeof(T), (size_t)RT * (64 * PT + 4) * 4) + 1023) / 1024 * 1024;
    const size_t lds = tile_b + (size_t)RT * C * sizeof(T);
    if (lds > (size_t)80 * 1024) return false;  // at least two blocks per CU
    if ((long)RT + (long)(taps - 1) * dil + 8 > CONV_HALO_SLACK) return false;
    const long gx = (e.M + RT - 1) / RT;
    const int xcd = conv_xcd_order();
    unsigned grid = 0;
    if (!conv_grid(gx, xcd, grid)) return false;
    // the kernels that read the f32 stream keep the tile's own rows as the residual: the stream the epilogue names must be that one
    if (in.Xf && (e.S_in != e.S_out || e.r != 1 || in.Xf != e.resid_f32 + ((long)e.P_out - (long)(taps - 1) * dil) * C)) return false;
    if constexpr (sizeof(T) == 2 && MT == 4 && (NT == 2 || NT == 3)) {
        static const bool plain = getenv("QTTS_UNIT_F32") && atoi(getenv("QTTS_UNIT_F32")) == 0;  // 0: the general kernel
        if (in.Xf && !plain && taps == 7 && (dil == 1 || dil == 3 || dil == 9) && unit_f32_epilogue(e, C) && (const void*)e.resid_f32 != (const void*)e.out_f32) {
            if (dil == 1) return launch_unit_f32<NT, 1>(st, W1, W2, wpk, bias1, alpha2, den2, e, in.Xf, in.alpha1, in.den1);
            if (dil == 3) return launch_unit_f32<NT, 3>(st, W1, W2, wpk, bias1, alpha2, den2, e, in.Xf, in.alpha1, in.den1);
            return launch_unit_f32<NT, 9>(st, W1, W2, wpk, bias1, alpha2, den2, e, in.Xf, in.alpha1, in.den1);
        }
    }
    if constexpr (sizeof(T) == 2 && MT == 4) {
        if (in.Xf) {
            auto* fnf = conv_unit_kernel<T, MT, NT, true>;
            QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fnf), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CEILING));
            hipLaunchKernelGGL(fnf, dim3(grid), dim3(256), lds, st, W1, W2, wpk, (const T*)nullptr, taps, dil, bias1, alpha2, den2, xcd, e, in.Xf, in.alpha1,
                               in.den1);
            return true;
        }
    }
    if (in.Xf) return false;
    auto* fn = conv_unit_kernel<T, MT, NT>;
    QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CEILING));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, xcd, e, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr);
    return true;
}

template <typename T>
bool try_unit(hipStream_t st, const T* W1, const T* W2, int wpk, const T* X, int C, int taps, int dil, const float* bias1, const float* alpha2,
              const float* den2, const ConvEpi& e, const UnitIn& in = UnitIn()) {
    if (e.N != C || e.r != 1 || !conv_unit_supported_fwd(sizeof(T) == 2 ? BF16 : F32, C, taps, dil)) return false;
    if (C == 64) return launch_unit<T, 1>(st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, e, in);
    if (C == 128) {
        // QTTS_UNIT_ROWS=128: 128-row tiles at 128 channels (every weight fragment feeds eight row tiles instead of four)
        static const int rows = getenv("QTTS_UNIT_ROWS") ? atoi(getenv("QTTS_UNIT_ROWS")) : 64;
        if constexpr (sizeof(T) == 2) {
            if (!in.Xf && rows == 128 && e.M >= 128L * 512 && launch_unit<T, 2, 8>(st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, e)) return true;
        }
        return launch_unit<T, 2>(st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, e, in);
    }
    if (C == 192) return launch_unit<T, 3>(st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, e, in);
    if constexpr (sizeof(T) == 2) {
        if (C == 256) return launch_unit<T, 4>(st, W1, W2, wpk, X, taps, dil, bias1, alpha2, den2, e, in);
    }
    return false;
}

template <typename T, int MT, int NT>
void launch_halo(hipStream_t st, const T* W, int wpk, const T* X, int K, int C, int taps, int dil, const ConvEpi& e, size_t lds) {
    const long RT = 16 * MT;
    const long gy = (e.M + RT - 1) / RT;
    const int xcd = conv_xcd_order();
    unsigned grid = 0;
    QTTS_CHECK(conv_grid(gy * (e.N / (64 * NT)), xcd, grid), ERR_SHAPE, "conv: too many tiles for one launch");
    auto* fn = conv_epi_stream_ok(e) ? conv_halo_kernel<T, MT, NT, true> : conv_halo_kernel<T, MT, NT, false>;
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, W, wpk, X, K, C, taps, dil, xcd, e);
}

template <typename T>
bool try_halo(hipStream_t st, const T* W, int wpk, const T* X, int K, int C, int taps, int dil, const ConvEpi& e) {
    // Tile shapes (rows x columns per block): 64 x 256 when N allows (four MFMAs per LDS fragment), 128 x 128 for
    // 128-column layers with many rows (halves the weight traffic), 64 x 128, else 128 x 64 / 64 x 64.
    const size_t row_bytes = (size_t)C * sizeof(T);
    const size_t halo = (size_t)(taps - 1) * dil;
    constexpr size_t LDS_MAX = 64 * 1024;
    const size_t otile64 = (size_t)64 * 132 * 4, otile128 = (size_t)128 * 68 * 4;
    const bool fits128 = std::max((128 + halo) * row_bytes, otile128) <= LDS_MAX;
    const bool fits64 = std::max((64 + halo) * row_bytes, otile64) <= LDS_MAX;
    const bool many = e.M >= 128L * 512;
    static const int tune = getenv("QTTS_CONV_TILE") ? atoi(getenv("QTTS_CONV_TILE")) : 0;  // bit 0: 64x256 tiles, bit 1: 128x128 tiles
    int mt, nt;
    if ((tune & 1) && e.N % 256 == 0 && fits64) { mt = 4; nt = 4; }
    else if ((tune & 2) && e.N % 128 == 0 && many && fits128) { mt = 8; nt = 2; }
    else if (e.N % 128 == 0 && fits64) { mt = 4; nt = 2; }
    else if (many && fits128) { mt = 8; nt = 1; }
    else if (fits64) { mt = 4; nt = 1; }
    else return false;
    const int rows = 16 * mt;
    if ((long)rows + (long)halo + 8 > CONV_HALO_SLACK) return false;  // the tile of the last block would read past the callers' slack
    const size_t lds = (std::max((rows + halo) * row_bytes, mt == 8 ? otile128 : otile64) + 1023) / 1024 * 1024;
    if ((e.M + rows - 1) / rows > 0x7fffffffL) return false;
    if (mt == 4 && nt == 4) launch_halo<T, 4, 4>(st, W, wpk, X, K, C, taps, dil, e, lds);
    else if (mt == 8 && nt == 2) launch_halo<T, 8, 2>(st, W, wpk, X, K, C, taps, dil, e, lds);
    else if (mt == 4 && nt == 2) launch_halo<T, 4, 2>(st, W, wpk, X, K, C, taps, dil, e, lds);
    else if (mt == 8) launch_halo<T, 8, 1>(st, W, wpk, X, K, C, taps, dil, e, lds);
    else launch_halo<T, 4, 1>(st, W, wpk, X, K, C, taps, dil, e, lds);
    return true;
}

}  // namespace

// Returns false when the shape does not fit the LDS-tile kernel (the caller then uses the tap-addressed GEMM).
// X rows must be contiguous (row stride == channels per tap) and readable for 256 rows + the halo past e.M.
bool launch_conv_halo(hipStream_t st, int dtype, WMat Wm, const void* X, int K, int ldx, int taps, long tap_stride, const ConvEpi& e, long x_lo) {
    const void* W = Wm.p;
    const int wpk = Wm.packed ? 1 : 0;
    if (taps < 1 || K % taps != 0 || !conv_epi_vector_ok(e)) return false;
    const int C = K / taps;
    const int chunk = dtype == BF16 ? 64 : 32;
    if (C != ldx || C % 64 != 0 || C % chunk != 0 || e.N % 64 != 0) return false;
    if (taps > 1 && (tap_stride <= 0 || tap_stride % ldx != 0)) return false;
    const long dil = taps > 1 ? tap_stride / ldx : 1;
    if (dil < 1 || dil > 64) return false;
    // QTTS_CONV_TILE_LO=0: inputs with a remainder plane through the tap-addressed GEMM, as before the wave-grid kernel walked them
    static const bool lo_off = getenv("QTTS_CONV_TILE_LO") && atoi(getenv("QTTS_CONV_TILE_LO")) == 0;
    if (x_lo != 0 && (dtype != BF16 || lo_off)) return false;
    if (dtype == BF16) {
        if (try_tile<bf16_t>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, K, C, taps, (int)dil, e, x_lo)) return true;
        if (x_lo != 0) return false;  // only the wave-grid kernel walks a remainder plane
        return try_halo<bf16_t>(st, (const bf16_t*)W, wpk, (const bf16_t*)X, K, C, taps, (int)dil, e);
    }
    if (try_tile<float>(st, (const float*)W, wpk, (const float*)X, K, C, taps, (int)dil, e)) return true;
    return try_halo<float>(st, (const float*)W, wpk, (const float*)X, K, C, taps, (int)dil, e);
}

// Shapes the fused residual unit covers: 64 / 128 / 192 (bf16: / 256) channels with a tile that leaves two blocks per CU.
bool conv_unit_supported(int dtype, int C, int taps, int dil) {
    static const bool off = getenv("QTTS_CONV_UNIT") && atoi(getenv("QTTS_CONV_UNIT")) == 0;
    if (off || taps < 1 || dil < 1 || dil > 64) return false;
    if (!(C == 64 || C == 128 || C == 192 || (C == 256 && dtype == BF16))) return false;
    const size_t es = dtype == BF16 ? 2 : 4;
    const int PT = (C / 64) % 2 == 0 ? 2 : 1;
    const size_t tile_b = (std::max((size_t)(64 + (taps - 1) * dil) * C * es, (size_t)64 * (64 * PT + 4) * 4) + 1023) / 1024 * 1024;
    return tile_b + (size_t)64 * C * es <= (size_t)76 * 1024;
}

namespace {
bool conv_unit_supported_fwd(int dtype, int C, int taps, int dil) { return conv_unit_supported(dtype, C, taps, dil); }
}  // namespace

// Fused residual unit: conv (taps, dilation; W1 [C][taps * C]) + bias1 + SnakeBeta(alpha2, den2), then the 1x1 conv W2 [C][C] with the
// epilogue e (bias, f32 residual, outputs).  X as for launch_conv_halo.  False = shape not covered (the caller launches the two
// convs one after the other).
bool launch_conv_unit(hipStream_t st, int dtype, WMat W1m, WMat W2m, const void* X, int C, int taps, int dil, const float* bias1,
                      const float* alpha2, const float* den2, const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1) {
    const void *W1 = W1m.p, *W2 = W2m.p;
    QTTS_CHECK(W1m.packed == W2m.packed, ERR_BAD_ARG, "conv_unit: the two weight matrices must share a layout");
    const int wpk = W1m.packed ? 1 : 0;
    QTTS_CHECK(W1 && W2 && (X || Xf) && bias1 && alpha2 && den2 && e.lens && e.M >= 1 && e.S_in >= 1, ERR_BAD_ARG, "conv_unit: null argument");
    if (!conv_epi_vector_ok(e)) return false;
    QTTS_CHECK(!Xf || (dtype == BF16 && alpha1 && den1 && !X), ERR_BAD_ARG, "conv_unit: an f32 input stream needs a bf16 run and the input activation's parameters");
    UnitIn in;
    in.Xf = Xf; in.alpha1 = alpha1; in.den1 = den1;
    if (dtype == BF16) return try_unit<bf16_t>(st, (const bf16_t*)W1, (const bf16_t*)W2, wpk, (const bf16_t*)X, C, taps, dil, bias1, alpha2, den2, e, in);
    return try_unit<float>(st, (const float*)W1, (const float*)W2, wpk, (const float*)X, C, taps, dil, bias1, alpha2, den2, e);
}

// ---------------------------------------------------------------------------------------------------------------------------
// The fused residual unit at 96 channels without the padding to 128 the general layout would give it (the last vocoder block of
// the 12 Hz decoder: 44 % of a padded unit's MFMAs and a quarter of its bytes are zeros).  Rows of the f32 stream are 96 floats.
// Weights come in a layout of their own (conv_unit96_pack): for every 16-column tile and every K step of 32 (7 taps x 3 steps for
// the dilated conv, 3 steps for the 1x1 conv) one KiB holding each lane's 16-byte A operand of the 16x16x32 MFMA.  Four waves as
// 2 x 2: row halves of the 64-row tile x three column tiles each.  LDS rows are pitched at 14 pieces of 16 bytes (224 B): a
// 16-lane group of a fragment read (eight rows of one piece, eight of the next, see tile_pcol) then covers all 64 banks without
// a swizzle, whatever the tap offset (at 13 pieces a third of the groups took two cycles).  Otherwise the persistent scheme of
// conv_unit_f32_kernel: halo reused inside LDS, own rows read once in the epilogue's thread layout (192 of the 256 threads: 24
// float4 per row), residual in registers, unrolled K walk with immediate LDS offsets.
// FIN: the unit is the decoder's last one and the output conv (96 -> 1, seven taps, causal, clamp to [-1, 1];
// decoder/v2.rs:238-246) runs on its activated output inside the kernel: the SnakeBeta copy goes to an LDS tile of 64 + 6 rows
// instead of to memory (its first six rows are the previous tile's last six; a workgroup starts one tile early, without storing,
// to have them for its first tile; rows without output are zeros, which is the causal padding in front of every strip), four
// threads per sample add up the 7 x 96 products, and only the waveform leaves the kernel.
template <int DIL, int PF, int BPC, bool PREFETCH, bool FIN = false>
__global__ __launch_bounds__(256, BPC) void conv_unit96_kernel(const uint4* __restrict__ W1, const uint4* __restrict__ W2,
                                                               const float* __restrict__ bias1, const float* __restrict__ alpha2,
                                                               const float* __restrict__ den2, int tiles_per_block, ConvEpi e,
                                                               const float* __restrict__ Xf, const float* __restrict__ alpha1,
                                                               const float* __restrict__ den1, Unit96Final fin) {
    using T = bf16_t;
    extern __shared__ uint4 tile[];
    constexpr int C = 96, RT = 64, HALO = 6 * DIL, ROWS_T = RT + HALO, PITCH = 14;  // pieces per LDS row
    constexpr int STEPS = 21, NT = 3, MT = 2, OP = C + 4, IT = 8;
    constexpr int tile_slots = (ROWS_T * PITCH + 63) / 64 * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const long total = (e.M + RT - 1) / RT;
    const long t0 = (long)blockIdx.x * tiles_per_block, t1 = t0 + tiles_per_block < total ? t0 + tiles_per_block : total;
    if (t0 >= total) return;
    const long ts = (FIN && t0 > 0) ? t0 - 1 : t0;  // first tile walked
    constexpr int YP = 13;                           // pieces per row of the activated output tile (12 + 1: rows 4 banks apart)
    __shared__ uint4 ytile[FIN ? (RT + 6) * YP : 1];
    __shared__ float fw[FIN ? 7 * C : 1];
    __shared__ long woff[FIN ? 2 : 1][FIN ? RT : 1];  // waveform element of every tile row (< 0: none)
    if constexpr (FIN) {
        for (int c = threadIdx.x; c < 7 * C; c += 256) fw[c] = fin.w[c];
        for (int c = threadIdx.x; c < 6 * YP; c += 256) ytile[c] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint4* htile = tile + tile_slots;
    float* otile = reinterpret_cast<float*>(htile);
    __shared__ int row_ok[2][RT];
    __shared__ float sn1[2 * C];
    // per-channel parameters of the tail (bias of the 7-tap conv, its SnakeBeta) and of the epilogue (bias of the 1x1 conv, the
    // SnakeBeta of the copy for the next layer), as snake_prep leaves them: read from memory once per workgroup instead of once
    // per tile, where each read was a round trip in front of the tail and of the epilogue
    __shared__ float prm[6 * C];
    for (int c = threadIdx.x; c < C; c += 256) {
        float al = alpha1[c], dn = den1[c];
        snake_prep<T>(al, dn);
        sn1[c] = al;
        sn1[C + c] = dn;
        prm[c] = bias1[c];
        al = alpha2[c]; dn = den2[c];
        snake_prep<T>(al, dn);
        prm[C + c] = al;
        prm[2 * C + c] = dn;
        prm[3 * C + c] = e.bias[c];
        al = 0.f; dn = 0.f;
        if (e.out2 || FIN) { al = e.alpha_exp[c]; dn = e.beta_den[c]; snake_prep<T>(al, dn); }
        prm[4 * C + c] = al;
        prm[5 * C + c] = dn;
    }
    const bool rowthr = threadIdx.x < 192;  // the threads that move rows: 24 float4 per row, eight rows per pass
    const int c4 = rowthr ? (int)threadIdx.x % 24 : 0, r0 = rowthr ? (int)threadIdx.x / 24 : 0;
    float res[IT][4], resn[PREFETCH ? IT : 1][4];
    // (threads 192 .. 255 read the first float4 of rows 0, 8, ... again instead of branching around the loads: with a branch in
    // between the compiler cannot count the loads in flight and every later wait becomes a wait for all of them)
    auto rows_load = [&](const float* src, float (&dst)[IT][4]) {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const float4 x = *reinterpret_cast<const float4*>(src + (long)(k * 8 + r0) * C + 4 * c4);
            dst[k][0] = x.x; dst[k][1] = x.y; dst[k][2] = x.z; dst[k][3] = x.w;
        }
    };
    auto snake4_store = [&](uint4* base, int row, int col, const float (&x)[4]) {
        const float4 al = *reinterpret_cast<const float4*>(sn1 + col), dn = *reinterpret_cast<const float4*>(sn1 + C + col);
        const float alv[4] = {al.x, al.y, al.z, al.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
        u16 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(snake_apply<T>(x[q], alv[q], dnv[q]));
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base + row * PITCH) + col * 2) =
            make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    };
    rows_load(Xf + (ts * RT + HALO) * C, res);
    __syncthreads();
    for (int idx = threadIdx.x; idx < HALO * 24; idx += 256) {  // the first tile's halo comes from memory
        const int row = idx / 24, col = 4 * (idx - row * 24);
        const float4 x = *reinterpret_cast<const float4*>(Xf + (ts * RT + row) * C + col);
        const float xv[4] = {x.x, x.y, x.z, x.w};
        snake4_store(tile, row, col, xv);
    }
    const uint4* lrow = tile + (32 * wr + r) * PITCH + g;   // the lane's piece of its row of every 16-row tile
    const uint4* hrow = htile + (32 * wr + r) * PITCH + g;
    for (long t = ts; t < t1; ++t) {
        const long m0 = t * RT;
        const bool mark = t == ts + 3;  // (time marks of a tile in the steady state: development probe builds only)
        if (mark) QTTS_TS(0);
        // parameters and weight addresses re-derived per tile: hoisted out of the loop they would sit in registers across the K walk
        long z = 0;  // (an opaque zero offset, not opaque pointers: see conv_unit_f32_kernel)
        asm volatile("" : "+s"(z));
        const uint4 *W1l = W1 + z, *W2l = W2 + z;
        int* ok = row_ok[t & 1];
        if ((int)threadIdx.x < RT) {
            const long m = m0 + threadIdx.x;
            int v = 0;
            long wo = -1;
            if (m < e.M) {
                const long u = m / e.S_in;
                const int tp = (int)(m - u * e.S_in);
                v = tp < e.lens[u];
                if (FIN && v && t >= t0) wo = u * fin.wav_stride + tp;
            }
            ok[threadIdx.x] = v;
            if constexpr (FIN) woff[t & 1][threadIdx.x] = wo;
        }
        if constexpr (!PREFETCH) {
            if (t > ts) rows_load(Xf + (m0 + HALO) * C, res);
        }
        const uint4* w1w = W1l + (size_t)(3 * wc) * STEPS * 64 + lane;  // column tile j, step s: + (j * STEPS + s) * 64
        bf16x8_t a[PF][NT];
#pragma unroll
        for (int s = 0; s < PF - 1; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j) a[s][j] = __builtin_bit_cast(bf16x8_t, w1w[(j * STEPS + s) * 64]);
        if (rowthr) {
#pragma unroll
            for (int k = 0; k < IT; ++k) snake4_store(tile, HALO + k * 8 + r0, 4 * c4, res[k]);
        }
        if constexpr (PREFETCH) rows_load(Xf + ((t + 1 < t1 ? m0 + RT : m0) + HALO) * C, resn);  // (the last tile reads its own rows again)
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (mark) QTTS_TS(1);
        // The K walk step by step, every step a scheduling region of its own: its weight loads for PF - 1 steps ahead, its LDS
        // reads for two steps ahead, then its MFMAs.  Without the region boundaries the scheduler sinks loads and reads next to
        // their uses, and every MFMA group then starts with a full LDS round trip.
        {
            constexpr int NPC = STEPS * MT, LDS_AHEAD = 2 * MT;
            bf16x8_t bq[NPC + LDS_AHEAD];
            auto frag = [&](int t) {
                const int s = t / MT, i = t % MT, tap = s / 3, ks = s % 3;
                return __builtin_bit_cast(bf16x8_t, lrow[(16 * i + tap * DIL) * PITCH + ks * 4]);
            };
#pragma unroll
            for (int t = 0; t < LDS_AHEAD; ++t) bq[t] = frag(t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s + PF - 1 < STEPS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) a[(s + PF - 1) % PF][j] = __builtin_bit_cast(bf16x8_t, w1w[(j * STEPS + QTTS_W_STEP(s + PF - 1)) * 64]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (s * MT + i + LDS_AHEAD < NPC) bq[s * MT + i + LDS_AHEAD] = frag(s * MT + i + LDS_AHEAD);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s % PF][j], bq[s * MT + i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (mark) QTTS_TS(2);
        // weights of the 1x1 conv: in flight while the tail of the 7-tap conv is computed
        bf16x8_t a2[3][NT];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j) a2[ks][j] = __builtin_bit_cast(bf16x8_t, W2l[((size_t)(3 * wc + j) * 3 + ks) * 64 + lane]);
        // tail of the 7-tap conv: bias, SnakeBeta of the bf16 value the unfused layer would have stored, into the second tile
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (3 * wc + j) * 16 + 4 * g;
            float bs[4], al[4], dn[4];
            ld4(prm + n, bs);
            ld4(prm + C + n, al);
            ld4(prm + 2 * C + n, dn);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = 32 * wr + 16 * i + r;
                u16 h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = bf16_bits_to_f32(f32_to_bf16_bits(acc[i][j][q] + bs[q]));
                    h[q] = f32_to_bf16_bits(snake_apply<T>(x, al[q], dn[q]));
                }
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(htile + row * PITCH) + n * 2) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();  // second tile complete; every wave is past its last read of the input tile
        if (mark) QTTS_TS(3);
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, hrow[16 * i * PITCH + ks * 4]);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ks][j], b, acc[i][j], 0, 0, 0);
            }
        if (mark) QTTS_TS(4);
        // the next tile's halo: rows [RT, RT + HALO) move to the front
        if (t + 1 < t1) {
            for (int sidx = threadIdx.x; sidx < HALO * PITCH; sidx += 256) tile[sidx] = tile[RT * PITCH + sidx];
        }
        __syncthreads();  // every wave is past its reads of the second tile: the f32 staging tile takes its place
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4_t*>(otile + (32 * wr + 16 * i + r) * OP + (3 * wc + j) * 16 + 4 * g) = acc[i][j];
        __syncthreads();
        if (mark) QTTS_TS(5);
        // The residual rows (and the next tile's) have been in flight since before the K walk.  One wait here, and the values
        // are passed through an empty asm statement: behind the conditional stores below the compiler can no longer count, and
        // would wait for everything in flight - the previous row's stores included - before every row.
        {
            const int col = 4 * c4;
            float b2[4], al[4], dn[4];
            ld4(prm + 3 * C + col, b2);
            ld4(prm + 4 * C + col, al);
            ld4(prm + 5 * C + col, dn);
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
#pragma unroll
            for (int k = 0; k < IT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    asm volatile("" : "+v"(res[k][q]));
                    if constexpr (PREFETCH) asm volatile("" : "+v"(resn[k][q]));
                }
          if (rowthr) {
            const long obase = (m0 + r0 + e.P_out) * C + col;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int row = k * 8 + r0;
                if constexpr (FIN) {
                    if (!ok[row]) {
                        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ytile + (6 + row) * YP) + col * 2) = make_uint2(0u, 0u);
                        continue;
                    }
                }
                if (!ok[row]) continue;
                const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + col);
                // (the sum replaces the residual in its registers: every row's store then reads registers of its own, and the
                // next row's LDS read need not wait until this row's store has fetched its data)
                float (&v)[4] = res[k];
                v[0] = a4.x + b2[0] + v[0]; v[1] = a4.y + b2[1] + v[1]; v[2] = a4.z + b2[2] + v[2]; v[3] = a4.w + b2[3] + v[3];
                const long off = obase + (long)k * 8 * C;
                if (e.out_f32) *reinterpret_cast<float4*>(e.out_f32 + off) = make_float4(v[0], v[1], v[2], v[3]);
                if constexpr (FIN) {
                    u16 h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(snake_apply<T>(v[q], al[q], dn[q]));
                    *reinterpret_cast<uint2*>(reinterpret_cast<char*>(ytile + (6 + row) * YP) + col * 2) =
                        make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                }
                if (e.out2) {
                    float y[4];
                    u16 h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = snake_apply<T>(v[q], al[q], dn[q]);
                        h[q] = f32_to_bf16_bits(y[q]);
                    }
                    T* o = (T*)e.out2 + off;
                    *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                    if (e.out2_lo) {
                        u16 l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) l[q] = f32_to_bf16_bits(y[q] - bf16_bits_to_f32(h[q]));
                        *reinterpret_cast<uint2*>(o + e.out2_lo) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    }
                }
            }
          }
        }
        if (mark) QTTS_TS(6);
        if constexpr (FIN) {
            __syncthreads();  // the tile's activated rows are in ytile behind the six carried over
            {
                const int sm = threadIdx.x >> 2, part = threadIdx.x & 3;  // sample, quarter of the channels
                float acc = 0.f;
#pragma unroll
                for (int kk = 0; kk < 7; ++kk) {
                    const bf16_t* yr = reinterpret_cast<const bf16_t*>(ytile + (sm + kk) * YP) + part * 24;
                    const float* wr = fw + kk * C + part * 24;
#pragma unroll
                    for (int c8 = 0; c8 < 3; ++c8) {
                        float yv[8], wv[8];
                        load8(yr + 8 * c8, yv);
                        load8(wr + 8 * c8, wv);
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc += yv[q] * wv[q];
                    }
                }
                acc += dpp_move<0xB1>(acc);  // quad_perm [1,0,3,2]
                acc += dpp_move<0x4E>(acc);  // quad_perm [2,3,0,1]
                const long wo = woff[t & 1][sm];
                if (part == 0 && wo >= 0) fin.wav[wo] = fminf(1.0f, fmaxf(-1.0f, acc + fin.bias));
            }
            __syncthreads();  // every sample has read its rows: the last six become the next tile's first six
            if ((int)threadIdx.x < 6 * YP) ytile[threadIdx.x] = ytile[RT * YP + threadIdx.x];
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int k = 0; k < IT; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) res[k][q] = resn[k][q];
        }
    }
}

// TWO consecutive 96-channel units (dilations DA, DB) in one persistent kernel: the first unit's output never goes to memory.  Its
// f32 values stay in the registers that held its residual and become the second unit's residual; their SnakeBeta copy goes into
// the second unit's LDS input tile, whose causal halo is, as for the first unit, the tail of the previous tile's rows.  A workgroup
// starts one tile early without storing (the second unit's halo of 6 DB rows must come from the first unit's output of the rows
// before the workgroup's own); rows without output pass zeros on, which is the causal padding in front of every strip (at least
// 6 * 9 such rows precede a strip's first row in the row space).  Every value is computed as by two launches of
// conv_unit96_kernel: the results are equal bit for bit.
struct Unit96Second {
    const uint4 *W1, *W2;                                // conv_unit96_pack layout
    const float *bias1, *alpha2, *den2, *bias2;          // 7-tap conv bias, its SnakeBeta, 1x1 conv bias
    const float *alpha1, *den1;                          // the second unit's input activation
};
template <int DA, int DB, int PF, int BPC>
__global__ __launch_bounds__(256, BPC) void conv_unit96x2_kernel(const uint4* __restrict__ W1, const uint4* __restrict__ W2,
                                                                 const float* __restrict__ bias1, const float* __restrict__ alpha2,
                                                                 const float* __restrict__ den2, int tiles_per_block, ConvEpi e,
                                                                 const float* __restrict__ Xf, const float* __restrict__ alpha1,
                                                                 const float* __restrict__ den1, Unit96Second sb) {
    using T = bf16_t;
    extern __shared__ uint4 tile[];
    constexpr int C = 96, RT = 64, HA = 6 * DA, HB = 6 * DB, PITCH = 14;
    constexpr int STEPS = 21, NT = 3, MT = 2, OP = C + 4, IT = 8;
    constexpr int slotsA = ((RT + HA) * PITCH + 63) / 64 * 64, slotsB = ((RT + HB) * PITCH + 63) / 64 * 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const long total = (e.M + RT - 1) / RT;
    const long t0 = (long)blockIdx.x * tiles_per_block, t1 = t0 + tiles_per_block < total ? t0 + tiles_per_block : total;
    if (t0 >= total) return;
    const long ts = t0 > 0 ? t0 - 1 : t0;  // first tile walked
    uint4* tileB = tile + slotsA;
    uint4* htile = tileB + slotsB;
    float* otile = reinterpret_cast<float*>(htile);
    __shared__ int row_ok[2][RT];
    __shared__ float snA[2 * C], snB[2 * C];
    __shared__ float prm[10 * C];  // unit A: bias1, alpha2', den2', bias2; unit B: the same four; the next layer's alpha', den'
    for (int c = threadIdx.x; c < C; c += 256) {
        float al = alpha1[c], dn = den1[c];
        snake_prep<T>(al, dn);
        snA[c] = al; snA[C + c] = dn;
        al = sb.alpha1[c]; dn = sb.den1[c];
        snake_prep<T>(al, dn);
        snB[c] = al; snB[C + c] = dn;
        prm[c] = bias1[c];
        al = alpha2[c]; dn = den2[c];
        snake_prep<T>(al, dn);
        prm[C + c] = al; prm[2 * C + c] = dn;
        prm[3 * C + c] = e.bias[c];
        prm[4 * C + c] = sb.bias1[c];
        al = sb.alpha2[c]; dn = sb.den2[c];
        snake_prep<T>(al, dn);
        prm[5 * C + c] = al; prm[6 * C + c] = dn;
        prm[7 * C + c] = sb.bias2[c];
        al = 0.f; dn = 0.f;
        if (e.out2) { al = e.alpha_exp[c]; dn = e.beta_den[c]; snake_prep<T>(al, dn); }
        prm[8 * C + c] = al; prm[9 * C + c] = dn;
    }
    for (int i = threadIdx.x; i < HB * PITCH; i += 256) tileB[i] = make_uint4(0u, 0u, 0u, 0u);  // (only the unstored first tile reads these)
    const bool rowthr = threadIdx.x < 192;
    const int c4 = rowthr ? (int)threadIdx.x % 24 : 0, r0 = rowthr ? (int)threadIdx.x / 24 : 0;
    float res[IT][4], resn[IT][4];
    auto rows_load = [&](const float* src, float (&dst)[IT][4]) {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const float4 x = *reinterpret_cast<const float4*>(src + (long)(k * 8 + r0) * C + 4 * c4);
            dst[k][0] = x.x; dst[k][1] = x.y; dst[k][2] = x.z; dst[k][3] = x.w;
        }
    };
    auto snake4_store = [&](uint4* base, const float* sn, int row, int col, const float (&x)[4]) {
        const float4 al = *reinterpret_cast<const float4*>(sn + col), dn = *reinterpret_cast<const float4*>(sn + C + col);
        const float alv[4] = {al.x, al.y, al.z, al.w}, dnv[4] = {dn.x, dn.y, dn.z, dn.w};
        u16 h[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(snake_apply<T>(x[q], alv[q], dnv[q]));
        *reinterpret_cast<uint2*>(reinterpret_cast<char*>(base + row * PITCH) + col * 2) =
            make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    };
    rows_load(Xf + (ts * RT + HA) * C, res);
    __syncthreads();
    for (int idx = threadIdx.x; idx < HA * 24; idx += 256) {  // the first tile's halo of the first unit comes from memory
        const int row = idx / 24, col = 4 * (idx - row * 24);
        const float4 x = *reinterpret_cast<const float4*>(Xf + (ts * RT + row) * C + col);
        const float xv[4] = {x.x, x.y, x.z, x.w};
        snake4_store(tile, snA, row, col, xv);
    }
    const uint4* lrowA = tile + (32 * wr + r) * PITCH + g;
    const uint4* lrowB = tileB + (32 * wr + r) * PITCH + g;
    const uint4* hrow = htile + (32 * wr + r) * PITCH + g;
    f32x4_t acc[MT][NT];
    // one unit's matrix work on an input tile: 7-tap conv, its tail into the second tile, 1x1 conv, accumulators into the staging tile
    auto unit_core = [&](auto dil_c, const uint4* lrow, uint4* in_tile, const uint4* W1l, const uint4* W2l, const float* pr) {
        constexpr int DIL = decltype(dil_c)::value;
        const uint4* w1w = W1l + (size_t)(3 * wc) * STEPS * 64 + lane;
        bf16x8_t a[PF][NT];
#pragma unroll
        for (int s = 0; s < PF - 1; ++s)
#pragma unroll
            for (int j = 0; j < NT; ++j) a[s][j] = __builtin_bit_cast(bf16x8_t, w1w[(j * STEPS + s) * 64]);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        __syncthreads();  // the input tile is complete
        {
            constexpr int NPC = STEPS * MT, LDS_AHEAD = 2 * MT;
            bf16x8_t bq[NPC + LDS_AHEAD];
            auto frag = [&](int t) {
                const int s = t / MT, i = t % MT, tap = s / 3, ks = s % 3;
                return __builtin_bit_cast(bf16x8_t, lrow[(16 * i + tap * DIL) * PITCH + ks * 4]);
            };
#pragma unroll
            for (int t = 0; t < LDS_AHEAD; ++t) bq[t] = frag(t);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                if (s + PF - 1 < STEPS) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) a[(s + PF - 1) % PF][j] = __builtin_bit_cast(bf16x8_t, w1w[(j * STEPS + s + PF - 1) * 64]);
                }
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    if (s * MT + i + LDS_AHEAD < NPC) bq[s * MT + i + LDS_AHEAD] = frag(s * MT + i + LDS_AHEAD);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[s % PF][j], bq[s * MT + i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bf16x8_t a2[3][NT];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int j = 0; j < NT; ++j) a2[ks][j] = __builtin_bit_cast(bf16x8_t, W2l[((size_t)(3 * wc + j) * 3 + ks) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = (3 * wc + j) * 16 + 4 * g;
            float bs[4], al[4], dn[4];
            ld4(pr + n, bs);
            ld4(pr + C + n, al);
            ld4(pr + 2 * C + n, dn);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int row = 32 * wr + 16 * i + r;
                u16 h[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = bf16_bits_to_f32(f32_to_bf16_bits(acc[i][j][q] + bs[q]));
                    h[q] = f32_to_bf16_bits(snake_apply<T>(x, al[q], dn[q]));
                }
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(htile + row * PITCH) + n * 2) =
                    make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();  // second tile complete; every wave is past its last read of the input tile
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const bf16x8_t b = __builtin_bit_cast(bf16x8_t, hrow[16 * i * PITCH + ks * 4]);
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[ks][j], b, acc[i][j], 0, 0, 0);
            }
        // the next tile's halo: rows [RT, RT + halo) move to the front
        for (int sidx = threadIdx.x; sidx < 6 * DIL * PITCH; sidx += 256) in_tile[sidx] = in_tile[RT * PITCH + sidx];
        __syncthreads();  // every wave is past its reads of the second tile: the f32 staging tile takes its place
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                *reinterpret_cast<f32x4_t*>(otile + (32 * wr + 16 * i + r) * OP + (3 * wc + j) * 16 + 4 * g) = acc[i][j];
        __syncthreads();
    };
    for (long t = ts; t < t1; ++t) {
        const long m0 = t * RT;
        long z = 0;  // (an opaque zero offset, not opaque pointers: see conv_unit_f32_kernel)
        asm volatile("" : "+s"(z));
        int* ok = row_ok[t & 1];
        if ((int)threadIdx.x < RT) {
            const long m = m0 + threadIdx.x;
            int v = 0;
            if (m < e.M) {
                const long u = m / e.S_in;
                v = (int)(m - u * e.S_in) < e.lens[u];
            }
            ok[threadIdx.x] = v;
        }
        if (rowthr) {
#pragma unroll
            for (int k = 0; k < IT; ++k) snake4_store(tile, snA, HA + k * 8 + r0, 4 * c4, res[k]);
        }
        rows_load(Xf + ((t + 1 < t1 ? m0 + RT : m0) + HA) * C, resn);  // (the last tile reads its own rows again)
        unit_core(std::integral_constant<int, DA>{}, lrowA, tile, W1 + z, W2 + z, prm);
        // ---- the first unit's epilogue: its sum replaces the residual in registers, its activated copy is the second unit's input
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the residual rows (and the next tile's) have landed, see conv_unit96_kernel
#pragma unroll
        for (int k = 0; k < IT; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(res[k][q]), "+v"(resn[k][q]));
        if (rowthr) {
            const int col = 4 * c4;
            float b2[4];
            ld4(prm + 3 * C + col, b2);
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int row = k * 8 + r0;
                float (&v)[4] = res[k];
                if (ok[row]) {
                    const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + col);
                    v[0] = a4.x + b2[0] + v[0]; v[1] = a4.y + b2[1] + v[1]; v[2] = a4.z + b2[2] + v[2]; v[3] = a4.w + b2[3] + v[3];
                } else {
                    v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f;  // SnakeBeta of 0 is 0
                }
                snake4_store(tileB, snB, HB + row, col, v);
            }
        }
        unit_core(std::integral_constant<int, DB>{}, lrowB, tileB, sb.W1 + z, sb.W2 + z, prm + 4 * C);
        // ---- the second unit's epilogue
        if (rowthr && t >= t0) {
            const int col = 4 * c4;
            float b2[4], al[4], dn[4];
            ld4(prm + 7 * C + col, b2);
            ld4(prm + 8 * C + col, al);
            ld4(prm + 9 * C + col, dn);
            const long obase = (m0 + r0 + e.P_out) * C + col;
#pragma unroll
            for (int k = 0; k < IT; ++k) {
                const int row = k * 8 + r0;
                if (!ok[row]) continue;
                const float4 a4 = *reinterpret_cast<const float4*>(otile + row * OP + col);
                float (&v)[4] = res[k];
                v[0] = a4.x + b2[0] + v[0]; v[1] = a4.y + b2[1] + v[1]; v[2] = a4.z + b2[2] + v[2]; v[3] = a4.w + b2[3] + v[3];
                const long off = obase + (long)k * 8 * C;
                if (e.out_f32) *reinterpret_cast<float4*>(e.out_f32 + off) = make_float4(v[0], v[1], v[2], v[3]);
                if (e.out2) {
                    float y[4];
                    u16 h[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        y[q] = snake_apply<T>(v[q], al[q], dn[q]);
                        h[q] = f32_to_bf16_bits(y[q]);
                    }
                    T* o = (T*)e.out2 + off;
                    *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
                    if (e.out2_lo) {
                        u16 l[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) l[q] = f32_to_bf16_bits(y[q] - bf16_bits_to_f32(h[q]));
                        *reinterpret_cast<uint2*>(o + e.out2_lo) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < IT; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) res[k][q] = resn[k][q];
    }
}

namespace {
template <int DIL, int PF, int BPC, bool PREFETCH>
bool launch_unit96_cfg(hipStream_t st, const void* W1, const void* W2, const float* bias1, const float* alpha2, const float* den2, const ConvEpi& e,
                       const float* Xf, const float* alpha1, const float* den1, const Unit96Final* fin) {
    constexpr int RT = 64, HALO = 6 * DIL, PITCH = 14;
    constexpr size_t in_b = (size_t)((RT + HALO) * PITCH + 63) / 64 * 64 * 16;
    constexpr size_t second = (size_t)RT * PITCH * 16, staging = (size_t)RT * (96 + 4) * 4;
    constexpr size_t lds = in_b + (second > staging ? second : staging);
    static_assert((lds + 22 * 1024) * BPC <= (size_t)156 * 1024, "unit tiles of the resident workgroups exceed the LDS of a CU");
    const long total = (e.M + RT - 1) / RT;
    if ((long)RT + HALO + 8 > CONV_HALO_SLACK || total > 0x7fffffffL) return false;
    const long blocks = std::min<long>(total, (long)BPC * 256);
    const int tpb = (int)((total + blocks - 1) / blocks);
    const unsigned grid = (unsigned)((total + tpb - 1) / tpb);
    auto* fn = fin ? conv_unit96_kernel<DIL, PF, BPC, PREFETCH, true> : conv_unit96_kernel<DIL, PF, BPC, PREFETCH, false>;
    // (the kernel with the output conv holds 20 KB of static LDS beside the dynamic tiles)
    QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, fin ? LDS_CEILING - 24 * 1024 : LDS_CEILING));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, (const uint4*)W1, (const uint4*)W2, bias1, alpha2, den2, tpb, e, Xf, alpha1, den1,
                       fin ? *fin : Unit96Final{});
    return true;
}
// Measured (13.7 M rows per launch, ms per unit): two workgroups per CU with prefetch 4.15 / 4.22 / 4.13 for dilations 1 / 3 / 9,
// three per CU without prefetch 4.12 / 4.29 / 4.04; the padded 128-channel unit takes 5.2 / 5.4 / 5.4 for the same rows.
template <int DIL>
bool launch_unit96_dil(hipStream_t st, const void* W1, const void* W2, const float* bias1, const float* alpha2, const float* den2, const ConvEpi& e,
                       const float* Xf, const float* alpha1, const float* den1, const Unit96Final* fin) {
    return launch_unit96_cfg<DIL, 4, 2, true>(st, W1, W2, bias1, alpha2, den2, e, Xf, alpha1, den1, fin);
}
}  // namespace

namespace {
template <int DA, int DB>
bool launch_unit96x2_cfg(hipStream_t st, const void* W1, const void* W2, const float* bias1, const float* alpha2, const float* den2, const ConvEpi& e,
                         const float* Xf, const float* alpha1, const float* den1, const Unit96Second& sb) {
    constexpr int RT = 64, PITCH = 14, BPC = 2;
    constexpr size_t inA = (size_t)((RT + 6 * DA) * PITCH + 63) / 64 * 64 * 16, inB = (size_t)((RT + 6 * DB) * PITCH + 63) / 64 * 64 * 16;
    constexpr size_t second = (size_t)RT * PITCH * 16, staging = (size_t)RT * (96 + 4) * 4;
    constexpr size_t lds = inA + inB + (second > staging ? second : staging);
    static_assert((lds + 7 * 1024) * BPC <= (size_t)156 * 1024, "unit tiles of the resident workgroups exceed the LDS of a CU");
    const long total = (e.M + RT - 1) / RT;
    if ((long)RT + 6 * DB + 8 > CONV_HALO_SLACK || total > 0x7fffffffL) return false;
    const long blocks = std::min<long>(total, (long)BPC * 256);
    const int tpb = (int)((total + blocks - 1) / blocks);
    const unsigned grid = (unsigned)((total + tpb - 1) / tpb);
    auto* fn = conv_unit96x2_kernel<DA, DB, 4, BPC>;
    QTTS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_CEILING - 8 * 1024));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, st, (const uint4*)W1, (const uint4*)W2, bias1, alpha2, den2, tpb, e, Xf, alpha1, den1, sb);
    return true;
}
}  // namespace

// Two consecutive units (dilations 1 and 3) in one launch: the first unit's arguments as for launch_conv_unit96 (e.bias its 1x1
// bias, e.resid_f32 / Xf the stream it reads), e's outputs are the pair's; the *b arguments belong to the second unit.  False:
// not covered.
bool launch_conv_unit96_pair(hipStream_t st, const void* W1, const void* W2, int dilA, int dilB, const float* bias1, const float* alpha2,
                             const float* den2, const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1, const void* W1b,
                             const void* W2b, const float* bias1b, const float* alpha2b, const float* den2b, const float* bias2b,
                             const float* alpha1b, const float* den1b) {
    QTTS_CHECK(W1 && W2 && Xf && bias1 && alpha2 && den2 && alpha1 && den1 && W1b && W2b && bias1b && alpha2b && den2b && bias2b && alpha1b && den1b &&
                   e.lens && e.M >= 1 && e.S_in >= 1, ERR_BAD_ARG, "conv_unit96_pair: null argument");
    if (!unit_f32_epilogue(e, 96) || e.N != 96 || (const void*)e.resid_f32 == (const void*)e.out_f32 || e.Cvalid != 0) return false;
    if (Xf != e.resid_f32 + ((long)e.P_out - 6L * dilA) * 96 || dilA != 1 || dilB != 3) return false;
    Unit96Second sb{(const uint4*)W1b, (const uint4*)W2b, bias1b, alpha2b, den2b, bias2b, alpha1b, den1b};
    return launch_unit96x2_cfg<1, 3>(st, W1, W2, bias1, alpha2, den2, e, Xf, alpha1, den1, sb);
}

// Weights of a 96-channel unit in the order conv_unit96_kernel reads them.  w: the conv's f32 weight (96, 96, taps) as the
// checkpoint holds it; out: 6 column tiles x (taps * 3) steps x 64 lanes x 8 bf16 - lane (r, g) of tile j, step (tap, ks) holds
// W[16 j + r][32 ks + 8 g .. + 7][tap].
void conv_unit96_pack(const float* w, int taps, uint16_t* out, uint16_t (*to_bf16)(float)) {
    const int steps = taps * 3;
    for (int j = 0; j < 6; ++j)
        for (int tap = 0; tap < taps; ++tap)
            for (int ks = 0; ks < 3; ++ks)
                for (int lane = 0; lane < 64; ++lane) {
                    const int r = lane & 15, g = lane >> 4, n = 16 * j + r;
                    uint16_t* dst = out + (((size_t)j * steps + tap * 3 + ks) * 64 + lane) * 8;
                    for (int el = 0; el < 8; ++el) dst[el] = to_bf16(w[((size_t)n * 96 + 32 * ks + 8 * g + el) * taps + tap]);
                }
}

// One residual unit at 96 unpadded channels on the f32 stream (bf16 matrix cores): e.N = e.Cout = e.ld_out = 96, e.resid_f32 = the
// input stream, Xf = that stream shifted to the tile's first halo row; W1 / W2 from conv_unit96_pack.  False: shape not covered.
// fin: the unit is the decoder's last and the output conv follows inside the kernel (e.out2 and e.out_f32 null, e.alpha_exp /
// e.beta_den the output conv's input activation; the waveform of strip u starts at fin->wav + u * fin->wav_stride).
bool launch_conv_unit96(hipStream_t st, const void* W1, const void* W2, int dil, const float* bias1, const float* alpha2, const float* den2,
                        const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1, const Unit96Final* fin) {
    QTTS_CHECK(W1 && W2 && Xf && bias1 && alpha2 && den2 && alpha1 && den1 && e.lens && e.M >= 1 && e.S_in >= 1, ERR_BAD_ARG, "conv_unit96: null argument");
    if (fin) {
        QTTS_CHECK(fin->w && fin->wav && !e.out2 && !e.out_f32 && e.alpha_exp && e.beta_den, ERR_BAD_ARG, "conv_unit96: bad output conv description");
        ConvEpi t = e;
        t.out2 = const_cast<float*>(Xf);  // (any non-null pointer: the shape check below wants an output)
        if (!unit_f32_epilogue(t, 96)) return false;
    } else if (!unit_f32_epilogue(e, 96)) {
        return false;
    }
    if (e.N != 96 || (const void*)e.resid_f32 == (const void*)e.out_f32 || e.Cvalid != 0) return false;
    if (Xf != e.resid_f32 + ((long)e.P_out - 6L * dil) * 96) return false;
    if (dil == 1) return launch_unit96_dil<1>(st, W1, W2, bias1, alpha2, den2, e, Xf, alpha1, den1, fin);
    if (dil == 3) return launch_unit96_dil<3>(st, W1, W2, bias1, alpha2, den2, e, Xf, alpha1, den1, fin);
    if (dil == 9) return launch_unit96_dil<9>(st, W1, W2, bias1, alpha2, den2, e, Xf, alpha1, den1, fin);
    return false;
}

}  // namespace qtts
