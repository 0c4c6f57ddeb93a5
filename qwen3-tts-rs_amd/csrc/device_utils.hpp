// Device-side helpers shared by the gfx950 kernels (wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>

// Development probe (tools/dev): with QTTS_KERNEL_TS defined, QTTS_TS(i) records the 100 MHz wall clock of the first thread of the
// launch's first and last block; product builds compile it to nothing.
#ifdef QTTS_KERNEL_TS
__device__ unsigned long long g_kernel_ts[2][16];
#define QTTS_TS(i)                                                                                                        \
    do {                                                                                                                  \
        if (threadIdx.x == 0 && blockIdx.y == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))                      \
            g_kernel_ts[blockIdx.x == 0 ? 0 : 1][i] = wall_clock64();                                                     \
    } while (0)
#else
#define QTTS_TS(i)
#endif

namespace qtts {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct bf16_t {  // storage-only 16-bit type so templates can switch on it
    u16 bits;
};

__device__ __forceinline__ float bf16_bits_to_f32(u16 b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ u16 f32_to_bf16_bits(float f) {
    __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN preserved
    return __builtin_bit_cast(u16, h);
}

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return bf16_bits_to_f32(v.bits); }

template <typename T>
__device__ __forceinline__ T from_f32(float v);
template <>
__device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return bf16_t{f32_to_bf16_bits(v)}; }

// load 8 consecutive elements as f32
__device__ __forceinline__ void load8(const float* p, float (&o)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p);
    const float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&o)[8]) {
    const uint4 a = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(a.x << 16); o[1] = __uint_as_float(a.x & 0xffff0000u);
    o[2] = __uint_as_float(a.y << 16); o[3] = __uint_as_float(a.y & 0xffff0000u);
    o[4] = __uint_as_float(a.z << 16); o[5] = __uint_as_float(a.z & 0xffff0000u);
    o[6] = __uint_as_float(a.w << 16); o[7] = __uint_as_float(a.w & 0xffff0000u);
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
    uint4 o;
    o.x = (unsigned)f32_to_bf16_bits(v[0]) | ((unsigned)f32_to_bf16_bits(v[1]) << 16);
    o.y = (unsigned)f32_to_bf16_bits(v[2]) | ((unsigned)f32_to_bf16_bits(v[3]) << 16);
    o.z = (unsigned)f32_to_bf16_bits(v[4]) | ((unsigned)f32_to_bf16_bits(v[5]) << 16);
    o.w = (unsigned)f32_to_bf16_bits(v[6]) | ((unsigned)f32_to_bf16_bits(v[7]) << 16);
    *reinterpret_cast<uint4*>(p) = o;
}

// Split activations.  A model-dtype activation tensor that feeds a GEMM may carry a second plane, `lo` elements behind the
// first, holding the rounding remainder T(v - f32(T(v))).  A GEMM that multiplies both planes sees v to about 16
// significant bits instead of bf16's 8, at bf16 MFMA rate (weights stay plain bf16).  lo == 0: single plane (always for f32).
template <typename T>
__device__ __forceinline__ void store_split(T* p, long lo, float v) {
    const T h = from_f32<T>(v);
    p[0] = h;
    if (lo) p[lo] = from_f32<T>(v - to_f32<T>(h));
}
// four consecutive elements at once (p and p + lo aligned to four elements): one 8- or 16-byte store per plane
template <typename T>
__device__ __forceinline__ void store4_split(T* p, long lo, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4_split<float>(float* p, long, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <>
__device__ __forceinline__ void store4_split<bf16_t>(bf16_t* p, long lo, const float (&v)[4]) {
    u16 h[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) h[q] = f32_to_bf16_bits(v[q]);
    *reinterpret_cast<uint2*>(p) = make_uint2((unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16));
    if (lo) {
        u16 l[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) l[q] = f32_to_bf16_bits(v[q] - bf16_bits_to_f32(h[q]));
        *reinterpret_cast<uint2*>(p + lo) = make_uint2((unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16));
    }
}
template <typename T>
__device__ __forceinline__ void store8_split(T* p, long lo, const float (&v)[8]);
template <>
__device__ __forceinline__ void store8_split<float>(float* p, long, const float (&v)[8]) { store8(p, v); }
template <>
__device__ __forceinline__ void store8_split<bf16_t>(bf16_t* p, long lo, const float (&v)[8]) {
    store8(p, v);
    if (lo) {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = v[e] - bf16_bits_to_f32(f32_to_bf16_bits(v[e]));
        store8(p + lo, r);
    }
}

// Sum of KS split-K partial slabs (element p[k * stride]) in slab order, with the loads of up to four slabs in flight
// at once (a plain loop serialises one memory round trip per slab).  The add order is the same as the plain loop's.
__device__ __forceinline__ float4 slab_sum4(const float* p, int KS, long stride) {
    float4 s = *reinterpret_cast<const float4*>(p);
    int k = 1;
    for (; k + 4 <= KS; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(p + (long)k * stride);
        const float4 b = *reinterpret_cast<const float4*>(p + (long)(k + 1) * stride);
        const float4 c = *reinterpret_cast<const float4*>(p + (long)(k + 2) * stride);
        const float4 d = *reinterpret_cast<const float4*>(p + (long)(k + 3) * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    if (k + 2 <= KS) {
        const float4 a = *reinterpret_cast<const float4*>(p + (long)k * stride);
        const float4 b = *reinterpret_cast<const float4*>(p + (long)(k + 1) * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        k += 2;
    }
    if (k < KS) {
        const float4 a = *reinterpret_cast<const float4*>(p + (long)k * stride);
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    return s;
}
// same for two scalars `gap` elements apart (a rotation pair of one head)
__device__ __forceinline__ void slab_sum_pair(const float* p, int gap, int KS, long stride, float& x1, float& x2) {
    float a1 = 0.f, a2 = 0.f;
    int k = 0;
    for (; k + 4 <= KS; k += 4) {
        const float* q0 = p + (long)k * stride;
        const float* q1 = q0 + stride;
        const float* q2 = q1 + stride;
        const float* q3 = q2 + stride;
        const float u0 = q0[0], v0 = q0[gap], u1 = q1[0], v1 = q1[gap], u2 = q2[0], v2 = q2[gap], u3 = q3[0], v3 = q3[gap];
        a1 += u0; a2 += v0;
        a1 += u1; a2 += v1;
        a1 += u2; a2 += v2;
        a1 += u3; a2 += v3;
    }
    for (; k < KS; ++k) {
        const float* q = p + (long)k * stride;
        a1 += q[0];
        a2 += q[gap];
    }
    x1 = a1;
    x2 = a2;
}

// ---- MFMA operand tiles.  A 16x16xK MFMA operand lane (r = lane & 15, g = lane >> 4) holds, of row r, the two 16-byte pieces
// 2g and 2g + 1 of every 128-byte row segment ("chunk": 64 bf16 / 32 f32 elements); piece h feeds the h-th MFMA step.
// Reading them straight from a row-major matrix makes 16 consecutive lanes touch 16 different cache lines, which a CU's
// vector L1 serves at about a quarter of its line rate (16 B/clk measured against 64 B/clk with eight consecutive lanes on one
// line), so:
//  - activations are DMA'd to LDS row-contiguously - instruction h2 of a 16-row x 128-byte unit covers rows 8 h2 .. 8 h2 + 7,
//    eight lanes per row - with the piece order inside a row XOR-swizzled so that the later ds_read_b128 fragment reads are
//    bank-conflict free (unit_src_* / unit_slot below);
//  - weights are re-laid once at load time into fragment order (see WMat in kernels.hpp), one contiguous KiB per wave load.
__device__ __forceinline__ int unit_swz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 2); }
// DMA instruction h2 (0/1) of a unit, this lane's source: row within the unit and 16-byte piece within the row segment
__device__ __forceinline__ int unit_src_row(int lane, int h2) { return 8 * h2 + (lane >> 3); }
__device__ __forceinline__ int unit_src_piece(int lane, int h2) { return (lane & 7) ^ unit_swz(unit_src_row(lane, h2)); }
// 16-byte LDS slot (0..127 within the unit's 2 KiB) of piece h of MFMA lane (r, g)
__device__ __forceinline__ int unit_slot(int r, int g, int h) { return r * 8 + ((2 * g + h) ^ unit_swz(r)); }

// Activations in MFMA fragment order ("packed"): a [64][N] bf16 matrix stored so that the B-operand load of a wave - row tile i,
// chunk c of 64 columns, half h - is one contiguous KiB: 16-byte piece index ((i N / 64 + c) 2 + h) 64 + lane, lane (r, g) holding
// columns 64 c + 8 (2 g + h) .. + 7 of row 16 i + r.  Element offset of (row m, column n):
__device__ __forceinline__ size_t packed_elem(int m, int n, int N) {
    const int pp = (n & 63) >> 3;
    return ((((size_t)(m >> 4) * (N >> 6) + (n >> 6)) * 2 + (pp & 1)) * 64 + (m & 15) + 16 * (pp >> 1)) * 8 + (n & 7);
}

// This lane's fragment of the 16-row weight tile that starts at row n0 (a multiple of 16) of W[N][K]: the piece pair of chunk c
// (128 bytes of a row) is at p + c * cs and p + c * cs + hs elements - in a row-major matrix, or in one re-laid into fragment
// order (pk != 0; layout: kernels.hpp WMat).
template <typename T>
struct WFragPtr {
    const T* p;
    int cs, hs;
    __device__ __forceinline__ WFragPtr() : p(nullptr), cs(0), hs(0) {}
    __device__ __forceinline__ WFragPtr(const T* W, int n0, int K, int lane, int pk) {
        constexpr int HALF = 16 / (int)sizeof(T), CHUNK = 8 * HALF, EPL = 2 * HALF;
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
    __device__ __forceinline__ const T* chunk(int c) const { return p + (size_t)c * cs; }
};

// Cross-lane reductions without LDS round trips.  The pairings are those of the __shfl_xor butterflies 32, 16, 8, 4, 2, 1 (every
// lane adds its partner's value, so the results are the same bit for bit): the gfx950 permlane swaps exchange 32-lane halves and
// 16-lane rows, DPP row rotations do the rest inside a row (after the step by 8 the values repeat every 8 lanes, so a rotation by
// 4 reaches the same partner value as lane ^ 4, and so on down).
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
constexpr int DPP_ROW_ROR = 0x120;  // + n: lane i of a 16-lane row reads lane (i + n) % 16
// a, b <- v: afterwards a holds (lo, lo) and b (hi, hi) of the 32-lane halves; op(a, b) is "own op partner" in every lane for a
// commutative op.  Written as inline assembly: the instruction updates both registers in place.
#define QTTS_SWAP_FOLD(NAME, INSN)                                            \
    template <typename F>                                                     \
    __device__ __forceinline__ float NAME(float v, F op) {                    \
        float a = v, b = v;                                                   \
        asm volatile("s_nop 1\n\t" INSN " %0, %1" : "+v"(a), "+v"(b));        \
        return op(a, b);                                                      \
    }
QTTS_SWAP_FOLD(fold_halves, "v_permlane32_swap_b32")
QTTS_SWAP_FOLD(fold_rows, "v_permlane16_swap_b32")
#undef QTTS_SWAP_FOLD
// sum over each 16-lane row, left in all its lanes (the butterfly 8, 4, 2, 1)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<DPP_ROW_ROR + 8>(v);
    v += dpp_move<DPP_ROW_ROR + 4>(v);
    v += dpp_move<DPP_ROW_ROR + 2>(v);
    v += dpp_move<DPP_ROW_ROR + 1>(v);
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
    const auto add = [](float x, float y) { return x + y; };
    v = fold_halves(v, add);
    v = fold_rows(v, add);
    return row16_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    const auto mx = [](float x, float y) { return fmaxf(x, y); };
    v = fold_halves(v, mx);
    v = fold_rows(v, mx);
    v = fmaxf(v, dpp_move<DPP_ROW_ROR + 8>(v));
    v = fmaxf(v, dpp_move<DPP_ROW_ROR + 4>(v));
    v = fmaxf(v, dpp_move<DPP_ROW_ROR + 2>(v));
    v = fmaxf(v, dpp_move<DPP_ROW_ROR + 1>(v));
    return v;
}

// Block-wide sum with a fixed combination order (deterministic). `red` holds >= 32 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

}  // namespace qtts
