// Decode GEMM at 64 token rows whose activations arrive in MFMA fragment order ("packed": the 16-byte piece (row tile i, chunk c,
// half h, lane) of a [64][K] bf16 matrix sits at piece index ((i nC + c) 2 + h) 64 + lane, nC = K / 64), so that a wave's B-operand
// load is one contiguous KiB straight from L2 to registers - no LDS staging of the X tile, no DMA, no barrier in front of the
// first MFMA.  A block of four waves owns NTB x MTB output tiles of 16 x 16; its waves take quarters of K, request every
// operand fragment of their quarter up front and sum their partial tiles through LDS in wave order.  What it is for: a small
// footprint (a few KB of LDS, 256 threads), so that the blocks of several decode streams share a CU instead of queueing for it.
#include "device_utils.hpp"
#include "kernels.hpp"

#include <cstdlib>

namespace qtts {

namespace {

struct GFrag {
    bf16x8_t v[2];
};
__device__ __forceinline__ f32x4_t gmma(const GFrag& a, const GFrag& b, f32x4_t c) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], b.v[1], c, 0, 0, 0);
    return c;
}
__device__ __forceinline__ GFrag frag_at(const bf16_t* base, size_t tile_chunk, int lane) {
    const bf16_t* p = base + tile_chunk * 1024 + lane * 8;
    GFrag f;
    f.v[0] = *reinterpret_cast<const bf16x8_t*>(p);
    f.v[1] = *reinterpret_cast<const bf16x8_t*>(p + 512);
    return f;
}

// MODE: FUSED_PLAIN (f32 out + bias), FUSED_SWIGLU (W rows interleaved [4 gate, 4 up, 4 gate, 4 up] per 16-row tile; act = silu(gate) up in
// packed order for a consumer with K = N / 2), FUSED_RESID (x = resid + sum + bias -> x_out f32; xb = bf16(x nw) in packed order; per-tile
// sums of squares).  PLAIN / SWIGLU multiply their finished sums by 1 / rms of the row from the producer's partial sums (e.ss_in), as
// the kernels of k_gemm_fused.hip do.
template <int NTB, int MTB, int CPW, int MODE, int NW>
__global__ __launch_bounds__(64 * NW) void gemm_frag_kernel(const bf16_t* __restrict__ W, const bf16_t* __restrict__ Xp, int N, int K, FusedEpi e) {
    __shared__ f32x4_t red[NW * NTB * MTB * 64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int nC = K / 64;
    const int ct0 = blockIdx.x * NTB, rt0 = blockIdx.y * MTB;
    const int c0 = wave * CPW;   // this wave's chunks [c0, c0 + CPW)
    GFrag wf[NTB][CPW], xf[MTB][CPW];
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int c = 0; c < CPW; ++c) wf[j][c] = frag_at(W, (size_t)(ct0 + j) * nC + c0 + c, lane);
#pragma unroll
    for (int i = 0; i < MTB; ++i)
#pragma unroll
        for (int c = 0; c < CPW; ++c) xf[i][c] = frag_at(Xp, (size_t)(rt0 + i) * nC + c0 + c, lane);
    // sums of squares of this block's 16 MTB rows from the producer's per-tile partials: eight groups of threads take every eighth
    // tile, so that the loads are in flight together (and beside the operand loads above)
    __shared__ float ssq[8][16 * MTB];
    if (MODE != FUSED_RESID && e.ss_in && threadIdx.x < 8 * 16 * MTB) {
        const int row = threadIdx.x % (16 * MTB), q = threadIdx.x / (16 * MTB);
        const float* p = e.ss_in + rt0 * 16 + row;
        float a0 = 0.f, a1 = 0.f;
        int t = q;
        for (; t + 8 < e.ntiles_in; t += 16) {
            const float v0 = p[(long)t * e.ss_ld], v1 = p[(long)(t + 8) * e.ss_ld];
            a0 += v0; a1 += v1;
        }
        for (; t < e.ntiles_in; t += 8) a0 += p[(long)t * e.ss_ld];
        ssq[q][row] = a0 + a1;
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4_t acc[NTB][MTB];
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int i = 0; i < MTB; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CPW; ++c)
#pragma unroll
        for (int j = 0; j < NTB; ++j)
#pragma unroll
            for (int i = 0; i < MTB; ++i) acc[j][i] = gmma(wf[j][c], xf[i][c], acc[j][i]);
#pragma unroll
    for (int j = 0; j < NTB; ++j)
#pragma unroll
        for (int i = 0; i < MTB; ++i) red[((wave * NTB + j) * MTB + i) * 64 + lane] = acc[j][i];
    __syncthreads();
    // tile t of the block is finished by wave t % 4
    for (int t = wave; t < NTB * MTB; t += NW) {
        const int j = t / MTB, i = t - j * MTB;
        f32x4_t v = red[((0 * NTB + j) * MTB + i) * 64 + lane];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[((w * NTB + j) * MTB + i) * 64 + lane];
        const int n = (ct0 + j) * 16 + 4 * g, m = (rt0 + i) * 16 + r;
        float x[4] = {v[0], v[1], v[2], v[3]};
        float rs = 1.0f;
        if (MODE != FUSED_RESID && e.ss_in) {
            const int lr = i * 16 + r;
            const float ss = ((ssq[0][lr] + ssq[1][lr]) + (ssq[2][lr] + ssq[3][lr])) + ((ssq[4][lr] + ssq[5][lr]) + (ssq[6][lr] + ssq[7][lr]));
            rs = 1.0f / sqrtf(ss / (float)e.H_in + e.eps);
        }
        if (MODE == FUSED_SWIGLU) {
            // lanes g = 0 / 2 hold gate values of four channels, the lanes 16 up the matching up values (read from the partial tiles)
            if ((g & 1) == 0) {
                f32x4_t u = red[((0 * NTB + j) * MTB + i) * 64 + lane + 16];
#pragma unroll
                for (int w = 1; w < NW; ++w) u += red[((w * NTB + j) * MTB + i) * 64 + lane + 16];
                float y[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float gv = x[q] * rs, uv = u[q] * rs;
                    y[q] = gv / (1.0f + expf(-gv)) * uv;
                }
                const int ch = (ct0 + j) * 8 + 4 * (g >> 1);  // first of four consecutive channels
                bf16_t* dst = (bf16_t*)e.act + packed_elem(m, ch, N / 2);
                *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)f32_to_bf16_bits(y[0]) | ((unsigned)f32_to_bf16_bits(y[1]) << 16),
                                                            (unsigned)f32_to_bf16_bits(y[2]) | ((unsigned)f32_to_bf16_bits(y[3]) << 16));
            }
            continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] *= rs;
        if (e.bias) {
            const float4 b4 = *reinterpret_cast<const float4*>(e.bias + n);
            x[0] += b4.x; x[1] += b4.y; x[2] += b4.z; x[3] += b4.w;
        }
        if (MODE == FUSED_PLAIN) {
            *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ld_out + n) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
            if (e.resid) {
                const float4 r4 = *reinterpret_cast<const float4*>(e.resid + (size_t)m * N + n);
                x[0] += r4.x; x[1] += r4.y; x[2] += r4.z; x[3] += r4.w;
            }
            *reinterpret_cast<float4*>(e.x_out + (size_t)m * N + n) = make_float4(x[0], x[1], x[2], x[3]);
            const float4 w4 = *reinterpret_cast<const float4*>(e.nw + n);
            bf16_t* dst = (bf16_t*)e.xb_out + packed_elem(m, n, N);
            const unsigned lo2 = (unsigned)f32_to_bf16_bits(x[0] * w4.x) | ((unsigned)f32_to_bf16_bits(x[1] * w4.y) << 16);
            const unsigned hi2 = (unsigned)f32_to_bf16_bits(x[2] * w4.z) | ((unsigned)f32_to_bf16_bits(x[3] * w4.w) << 16);
            *reinterpret_cast<uint2*>(dst) = make_uint2(lo2, hi2);
            float ss = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            const auto add = [](float x0, float x1) { return x0 + x1; };
            ss = fold_rows(ss, add);
            ss = fold_halves(ss, add);
            if (g == 0) e.ss_out[(long)(ct0 + j) * e.ss_ld + m] = ss;
        }
    }
}

template <int NTB, int MTB, int CPW, int NW = 4>
void launch_frag_mode(hipStream_t st, const bf16_t* W, const bf16_t* Xp, int N, int K, const FusedEpi& e, int rows) {
    dim3 grid(N / (16 * NTB), rows / 16 / MTB);
    if (e.mode == FUSED_PLAIN) hipLaunchKernelGGL((gemm_frag_kernel<NTB, MTB, CPW, FUSED_PLAIN, NW>), grid, dim3(64 * NW), 0, st, W, Xp, N, K, e);
    else if (e.mode == FUSED_SWIGLU) hipLaunchKernelGGL((gemm_frag_kernel<NTB, MTB, CPW, FUSED_SWIGLU, NW>), grid, dim3(64 * NW), 0, st, W, Xp, N, K, e);
    else hipLaunchKernelGGL((gemm_frag_kernel<NTB, MTB, CPW, FUSED_RESID, NW>), grid, dim3(64 * NW), 0, st, W, Xp, N, K, e);
}

}  // namespace

bool gemm_frag_covers(int N, int K, int mode) {
    if (K % 256 != 0 || N % 64 != 0) return false;
    const int cpw = K / 64 / 4;
    if (mode == FUSED_RESID || N <= 2048) return cpw == 2 || cpw == 4 || cpw == 8 || cpw == 12;
    return cpw == 4 || cpw == 8;
}

// rows = 64 or 128; W fragment order [N][K]; Xp packed [rows][K].  16 x 16 output tiles per block for N <= 2048 and for RESID (N = hidden,
// long K), 32 x 32 otherwise.
bool launch_gemm_frag(hipStream_t st, WMat Wm, const void* Xp, int N, int K, const FusedEpi& e, int rows) {
    if (!Wm.packed || !gemm_frag_covers(N, K, e.mode) || (rows != 64 && rows != 128)) return false;
    QTTS_CHECK(e.mode != FUSED_PLAIN || (e.out_f32 && e.ld_out >= N), ERR_BAD_ARG, "gemm_frag: missing output");
    QTTS_CHECK(e.mode != FUSED_SWIGLU || e.act, ERR_BAD_ARG, "gemm_frag: missing activation output");
    QTTS_CHECK(e.mode != FUSED_RESID || (e.x_out && e.xb_out && e.nw && e.ss_out && e.ss_ld >= rows), ERR_BAD_ARG, "gemm_frag: missing residual outputs");
    QTTS_CHECK(!e.ss_in || (e.ntiles_in >= 1 && e.H_in >= 1 && e.ss_ld >= rows), ERR_BAD_ARG, "gemm_frag: bad norm partials");
    const bf16_t* w = (const bf16_t*)Wm.p;
    const bf16_t* x = (const bf16_t*)Xp;
    const int cpw = K / 64 / 4;
    static const int nw8 = getenv("QTTS_FRAG_WAVES") ? atoi(getenv("QTTS_FRAG_WAVES")) == 8 : 0;
    if (e.mode == FUSED_RESID || N <= 2048) {
        if (nw8 && cpw == 8) launch_frag_mode<1, 1, 4, 8>(st, w, x, N, K, e, rows);
        else if (nw8 && cpw == 12) launch_frag_mode<1, 1, 6, 8>(st, w, x, N, K, e, rows);
        else if (cpw == 8) launch_frag_mode<1, 1, 8>(st, w, x, N, K, e, rows);
        else if (cpw == 12) launch_frag_mode<1, 1, 12>(st, w, x, N, K, e, rows);
        else if (cpw == 4) launch_frag_mode<1, 1, 4>(st, w, x, N, K, e, rows);
        else if (cpw == 2) launch_frag_mode<1, 1, 2>(st, w, x, N, K, e, rows);
        else return false;
    } else {
        if (cpw == 4) launch_frag_mode<2, 2, 4>(st, w, x, N, K, e, rows);
        else if (cpw == 8) launch_frag_mode<2, 2, 8>(st, w, x, N, K, e, rows);
        else return false;
    }
    return true;
}

}  // namespace qtts
