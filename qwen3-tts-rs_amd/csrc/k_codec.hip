// Codec-decoder kernels around the large-M GEMM (qwen_tts/src/audio/decoder/v2.rs:163-247).
//
// Activations are channels-last, one padded strip per utterance: strip u occupies rows
// [u*S, (u+1)*S) with P leading zero rows (the causal left padding of every conv that reads the
// tensor, audio/tokenizer/v2/causal_conv.rs:91-101) followed by T time steps.  With that layout a
// causal conv is a GEMM whose taps are row offsets, and a transposed conv with stride r is a GEMM with
// N = r * Cout whose output rows are already the r-times longer sequence.
#include "conv_epi.hpp"

namespace qtts {

// ------------------------------------------------------------------------------------------------
// SplitResidualVectorQuantizer::decode (audio/quantizer.rs:417-434) with each group's 1x1 output_proj
// folded into its codebooks at load time: x[u][t] = sum_q table[q][code[u][t][q]].
template <typename T>
__global__ __launch_bounds__(256) void rvq_gather_kernel(const float* __restrict__ table, const unsigned* __restrict__ codes,
                                                         const int* __restrict__ rows, const int* __restrict__ starts,
                                                         const int* __restrict__ lens, int Q, int csize, int Dc, int T_max,
                                                         long S, int P, T* out, int ld, long out_lo) {
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    const unsigned* c = codes + ((long)rows[u] * T_max + starts[u] + t) * Q;
    T* o = out + ((long)u * S + P + t) * ld;
    for (int i = threadIdx.x; i < Dc; i += blockDim.x) {
        float s = table[((long)0 * csize + min(c[0], (unsigned)csize - 1)) * Dc + i];
        for (int q = 1; q < Q; ++q) s = s + table[((long)q * csize + min(c[q], (unsigned)csize - 1)) * Dc + i];
        store_split<T>(o + i, out_lo, s);
    }
}

void launch_rvq_gather(hipStream_t st, int dtype, const float* table, const unsigned* codes, const int* rows, const int* starts,
                       const int* lens, int U, int Tc, int Q, int csize, int Dc, int T_max, long S, int P, void* out, int ld, long out_lo) {
    QTTS_CHECK(U >= 1 && Tc >= 1 && Dc <= ld, ERR_SHAPE, "rvq_gather: bad shape");
    QTTS_CHECK(out_lo == 0 || dtype == BF16, ERR_BAD_ARG, "rvq_gather: a remainder plane exists for bf16 activations only");
    dim3 grid(Tc, U);
    if (dtype == BF16) hipLaunchKernelGGL(rvq_gather_kernel<bf16_t>, grid, dim3(256), 0, st, table, codes, rows, starts, lens, Q, csize, Dc, T_max, S, P, (bf16_t*)out, ld, out_lo);
    else hipLaunchKernelGGL(rvq_gather_kernel<float>, grid, dim3(256), 0, st, table, codes, rows, starts, lens, Q, csize, Dc, T_max, S, P, (float*)out, ld, 0L);
}

// ------------------------------------------------------------------------------------------------
// Epilogue of every conv GEMM: bias, optional tanh-GELU (ConvNeXt pwconv1, audio/tokenizer/v2/convnext.rs:81),
// optional per-channel scale (ConvNeXt gamma), optional residual, then the raw result and/or its
// SnakeBeta (audio/tokenizer/v2/snake_beta.rs:37-73: x + sin^2(x e^alpha) / (e^beta + 1e-9), in f32).
template <typename T>
__global__ __launch_bounds__(256) void conv_epilogue_kernel(ConvEpi e) {
    const long total = e.M * (long)e.N / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long m = idx * 4 / e.N;
        const int n = (int)(idx * 4 % e.N);
        const long u = m / e.S_in;
        const int tp = (int)(m % e.S_in);
        if (tp >= e.lens[u]) continue;  // rows past the strip's last valid step only hold padding / the next strip
        const float4 a = *reinterpret_cast<const float4*>(e.acc + m * e.N + n);
        float v[4] = {a.x, a.y, a.z, a.w};
        conv_epi_apply<T>(e, u, tp, n, v);
    }
}

void launch_conv_epilogue(hipStream_t st, int dtype, const ConvEpi& e) {
    QTTS_CHECK(e.N % 4 == 0 && e.Cout % 4 == 0 && e.N == e.r * e.Cout && e.M >= 1 && e.Cvalid % 4 == 0 && conv_epi_vector_ok(e), ERR_SHAPE, "conv_epilogue: bad shape");
    QTTS_CHECK(e.out2 == nullptr || e.act2 == 1 || (e.alpha_exp && e.beta_den), ERR_BAD_ARG, "conv_epilogue: snake output without parameters");
    QTTS_CHECK(e.acc != nullptr, ERR_BAD_ARG, "conv_epilogue: GEMM output missing");
    const long total = e.M * (long)e.N / 4;
    const int blocks = (int)std::min<long>((total + 255) / 256, 8192);
    if (dtype == BF16) hipLaunchKernelGGL(conv_epilogue_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, e);
    else hipLaunchKernelGGL(conv_epilogue_kernel<float>, dim3(blocks), dim3(256), 0, st, e);
}

// ------------------------------------------------------------------------------------------------
// ConvNeXt front (convnext.rs:69-79): depthwise causal conv k=7 (+bias) then LayerNorm(eps 1e-6) over channels.
template <typename T>
__global__ __launch_bounds__(256) void dwconv_ln_kernel(const T* __restrict__ x, const float* __restrict__ w /*[C][7]*/,
                                                        const float* __restrict__ b, const float* __restrict__ ln_w,
                                                        const float* __restrict__ ln_b, const int* __restrict__ lens,
                                                        long S, int P, int C, T* out, long x_lo) {
    __shared__ float red[32];
    extern __shared__ float ybuf[];
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    const long row = (long)u * S + P + t;
    float sum = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = b[c];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            float xv = to_f32<T>(x[(row + k - 6) * C + c]);
            if (x_lo) xv += to_f32<T>(x[x_lo + (row + k - 6) * C + c]);
            acc += w[c * 7 + k] * xv;
        }
        ybuf[c] = acc;
        sum += acc;
    }
    const float mean = block_sum(sum, red) / (float)C;
    float var = 0.f;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float d = ybuf[c] - mean;
        var += d * d;
    }
    var = block_sum(var, red) / (float)C;
    const float den = sqrtf(var + 1e-6f);
    for (int c = threadIdx.x; c < C; c += blockDim.x) out[row * C + c] = from_f32<T>((ybuf[c] - mean) / den * ln_w[c] + ln_b[c]);
}

void launch_dwconv_ln(hipStream_t st, int dtype, const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b,
                      const int* lens, int U, int Tmax, long S, int P, int C, void* out, long x_lo) {
    QTTS_CHECK(P >= 6 && C >= 1 && C * 4 <= 64 * 1024, ERR_SHAPE, "dwconv_ln: needs 6 padding rows and C <= 16384");
    QTTS_CHECK(x_lo == 0 || dtype == BF16, ERR_BAD_ARG, "dwconv_ln: a remainder plane exists for bf16 activations only");
    dim3 grid(Tmax, U);
    if (dtype == BF16) hipLaunchKernelGGL(dwconv_ln_kernel<bf16_t>, grid, dim3(256), (size_t)C * 4, st, (const bf16_t*)x, w, b, ln_w, ln_b, lens, S, P, C, (bf16_t*)out, x_lo);
    else hipLaunchKernelGGL(dwconv_ln_kernel<float>, grid, dim3(256), (size_t)C * 4, st, (const float*)x, w, b, ln_w, ln_b, lens, S, P, C, (float*)out, 0L);
}

// ------------------------------------------------------------------------------------------------
// Final conv (C -> 1, k = 7, causal) + clamp[-1, 1] (decoder/v2.rs:238-246).  A block produces 128 consecutive samples: their
// 134 input rows are staged in LDS with 16-byte loads.  A thread computes FOUR consecutive samples over an eighth of the
// channels, so a weight read from LDS feeds four samples and an input row up to four taps; eight lanes share a sample group.
// (64 samples per block in f32 runs, whose rows are twice as wide)
template <typename T>
__global__ __launch_bounds__(256) void final_conv_kernel(const T* __restrict__ x, const float* __restrict__ w /*[7][C]*/, float bias,
                                                         const int* __restrict__ nvalid, long S, int P, int C, float* wav,
                                                         long wav_stride) {
    constexpr int FC_SAMPLES = sizeof(T) == 2 ? 128 : 64, FC_ROWS = FC_SAMPLES + 6;
    extern __shared__ __align__(16) unsigned char smem[];
    const int u = blockIdx.y;
    const long t0 = (long)blockIdx.x * FC_SAMPLES;
    const int n = nvalid[u];
    if (t0 >= n) return;
    const int row_bytes = C * (int)sizeof(T), pitch = row_bytes + 16;  // +16: consecutive rows start 4 banks apart
    const int vpr = row_bytes / 16;
    float* wl = reinterpret_cast<float*>(smem + FC_ROWS * pitch);
    const unsigned char* src = reinterpret_cast<const unsigned char*>(x + ((long)u * S + P + t0 - 6) * C);
    // four loads in flight per thread before the first LDS store (a load-store-load chain would pay the memory latency per vector)
    for (int base = 0; base < FC_ROWS * vpr; base += 1024) {
        uint4 buf[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = base + k * 256 + (int)threadIdx.x;
            if (v < FC_ROWS * vpr) buf[k] = *reinterpret_cast<const uint4*>(src + (long)(v / vpr) * row_bytes + (v % vpr) * 16);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = base + k * 256 + (int)threadIdx.x;
            if (v < FC_ROWS * vpr) *reinterpret_cast<uint4*>(smem + (v / vpr) * pitch + (v % vpr) * 16) = buf[k];
        }
    }
    for (int i = threadIdx.x; i < 7 * C; i += 256) wl[i] = w[i];
    __syncthreads();
    const int sg = threadIdx.x >> 3, slice = threadIdx.x & 7;
    if (4 * sg >= FC_SAMPLES) return;
    // channels of a slice: a contiguous eighth where that is a whole number of 8-channel pieces, else every eighth piece
    const bool contiguous = C % 64 == 0;
    const int cs = contiguous ? C / 8 : C, c0 = contiguous ? slice * cs : slice * 8, cstep = contiguous ? 8 : 64;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c0 + c < (contiguous ? c0 + cs : C); c += cstep) {
        float wk[7][8];
#pragma unroll
        for (int k = 0; k < 7; ++k) load8(wl + k * C + c0 + c, wk[k]);
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            float xv[8];
            load8(reinterpret_cast<const T*>(smem + (4 * sg + r) * pitch) + c0 + c, xv);
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int k = r - sidx;  // sample 4 sg + sidx reads rows 4 sg + sidx .. + 6
                if (k >= 0 && k <= 6) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc[sidx] += xv[q] * wk[k][q];
                }
            }
        }
    }
#pragma unroll
    for (int sidx = 0; sidx < 4; ++sidx) {
        acc[sidx] += __shfl_xor(acc[sidx], 1, 64);
        acc[sidx] += __shfl_xor(acc[sidx], 2, 64);
        acc[sidx] += __shfl_xor(acc[sidx], 4, 64);
    }
    if (slice == 0) {
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const long t = t0 + 4 * sg + sidx;
            if (t < n) wav[(long)u * wav_stride + t] = fminf(1.0f, fmaxf(-1.0f, acc[sidx] + bias));
        }
    }
}

void launch_final_conv(hipStream_t st, int dtype, const void* x, const float* w, float bias, const int* nvalid, int U, long Tmax, long S,
                       int P, int C, float* wav, long wav_stride) {
    QTTS_CHECK(P >= 6 && C % 8 == 0 && C >= 8 && U >= 1 && Tmax >= 1, ERR_SHAPE, "final_conv: needs six padding rows and a channel count that is a multiple of 8");
    const size_t es = dtype == BF16 ? 2 : 4;
    const int per_block = dtype == BF16 ? 128 : 64;
    const size_t lds = (size_t)(per_block + 6) * ((size_t)C * es + 16) + (size_t)7 * C * 4;
    QTTS_CHECK(lds <= 64 * 1024, ERR_SHAPE, "final_conv: too many channels for the LDS staging buffer");
    dim3 grid((unsigned)((Tmax + per_block - 1) / per_block), U);
    if (dtype == BF16) hipLaunchKernelGGL(final_conv_kernel<bf16_t>, grid, dim3(256), lds, st, (const bf16_t*)x, w, bias, nvalid, S, P, C, wav, wav_stride);
    else hipLaunchKernelGGL(final_conv_kernel<float>, grid, dim3(256), lds, st, (const float*)x, w, bias, nvalid, S, P, C, wav, wav_stride);
}

}  // namespace qtts
