// Kernels of the codec ENCODER (waveform -> codes) that are not GEMMs
// (qwen_tts/src/audio/encoder/{seanet,transformer,quantizer}.rs).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// SeaNet input conv (seanet.rs:102,155-156): 1 -> 64 channels, k = 7, causal (6 leading zeros).  Writes the raw output
// (residual source of the first block) and its ELU (input of the block's first conv) as channels-last strips.
template <typename T>
__global__ __launch_bounds__(256) void enc_init_conv_kernel(const float* __restrict__ audio, const int* __restrict__ n_samples, long max_samples,
                                                            const float* __restrict__ w /*[64][7]*/, const float* __restrict__ bias, long S, int P,
                                                            T* out_raw, T* out_elu) {
    __shared__ float xs[64 + 6];
    __shared__ float ws[64 * 7 + 64];
    const int u = blockIdx.y;
    const long t0 = (long)blockIdx.x * 64;
    const int n = n_samples[u];
    if (t0 >= n) return;
    for (int i = threadIdx.x; i < 70; i += 256) {
        const long s = t0 + i - 6;
        xs[i] = (s >= 0 && s < n) ? audio[(long)u * max_samples + s] : 0.f;
    }
    for (int i = threadIdx.x; i < 64 * 7; i += 256) ws[i] = w[i];
    if (threadIdx.x < 64) ws[64 * 7 + threadIdx.x] = bias[threadIdx.x];
    __syncthreads();
    const int c = threadIdx.x & 63;
    for (int tt = threadIdx.x >> 6; tt < 64; tt += 4) {
        const long t = t0 + tt;
        if (t >= n) break;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) acc += ws[c * 7 + j] * xs[tt + j];
        acc += ws[64 * 7 + c];
        const long o = ((long)u * S + P + t) * 64 + c;
        const T r = from_f32<T>(acc);
        out_raw[o] = r;
        const float x = to_f32<T>(r);
        out_elu[o] = from_f32<T>(x > 0.f ? x : expf(x) - 1.0f);
    }
}
void launch_enc_init_conv(hipStream_t st, int dtype, const float* audio, const int* n_samples, int U, long max_samples, const float* w,
                          const float* bias, long S, int P, void* out_raw, void* out_elu) {
    QTTS_CHECK(U >= 1 && max_samples >= 1 && S >= P + max_samples, ERR_SHAPE, "enc_init_conv: bad shape");
    dim3 grid((unsigned)((max_samples + 63) / 64), U);
    if (dtype == BF16) hipLaunchKernelGGL(enc_init_conv_kernel<bf16_t>, grid, dim3(256), 0, st, audio, n_samples, max_samples, w, bias, S, P, (bf16_t*)out_raw, (bf16_t*)out_elu);
    else hipLaunchKernelGGL(enc_init_conv_kernel<float>, grid, dim3(256), 0, st, audio, n_samples, max_samples, w, bias, S, P, (float*)out_raw, (float*)out_elu);
}

// Residual + LayerScale + LayerNorm of the encoder transformer (transformer.rs:170-184; candle layer_norm, eps 1e-5).
template <typename T>
__global__ __launch_bounds__(256) void resid_layernorm_kernel(const float* __restrict__ resid, const float* __restrict__ part, int KS, long ks_stride,
                                                              const float* __restrict__ scale, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, float eps, float* resid_out, T* xn, int H) {
    __shared__ float red[32];
    const long row = blockIdx.x;
    constexpr int RMAX = 2;  // rows up to 2048 wide
    float4 keep[RMAX];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < RMAX; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (i < H) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (part) s = slab_sum4(part + row * H + i, KS, ks_stride);
            if (scale) {
                const float4 c = *reinterpret_cast<const float4*>(scale + i);
                s.x *= c.x; s.y *= c.y; s.z *= c.z; s.w *= c.w;
            }
            if (resid) {
                const float4 r = *reinterpret_cast<const float4*>(resid + row * H + i);
                s.x = r.x + s.x; s.y = r.y + s.y; s.z = r.z + s.z; s.w = r.w + s.w;
            }
            *reinterpret_cast<float4*>(resid_out + row * H + i) = s;
            keep[it] = s;
            sum += s.x + s.y + s.z + s.w;
        }
    }
    if (!ln_w) return;
    const float mean = block_sum(sum, red) / (float)H;
    float var = 0.f;
#pragma unroll
    for (int it = 0; it < RMAX; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (i < H) {
            const float4 x = keep[it];
            const float a = x.x - mean, b = x.y - mean, c = x.z - mean, d = x.w - mean;
            var += a * a + b * b + c * c + d * d;
        }
    }
    const float den = sqrtf(block_sum(var, red) / (float)H + eps);
#pragma unroll
    for (int it = 0; it < RMAX; ++it) {
        const int i = threadIdx.x * 4 + it * 1024;
        if (i < H) {
            const float4 x = keep[it];
            const float4 g = *reinterpret_cast<const float4*>(ln_w + i);
            const float4 bb = *reinterpret_cast<const float4*>(ln_b + i);
            T* o = xn + row * H + i;
            o[0] = from_f32<T>((x.x - mean) / den * g.x + bb.x);
            o[1] = from_f32<T>((x.y - mean) / den * g.y + bb.y);
            o[2] = from_f32<T>((x.z - mean) / den * g.z + bb.z);
            o[3] = from_f32<T>((x.w - mean) / den * g.w + bb.w);
        }
    }
}
void launch_resid_layernorm(hipStream_t st, int dtype, const float* resid, const float* part, int KS, long ks_stride, const float* scale,
                            const float* ln_w, const float* ln_b, float eps, float* resid_out, void* xn, int M, int H) {
    QTTS_CHECK(M >= 1 && H % 4 == 0 && H <= 2048 && resid_out && (ln_w == nullptr || (ln_b && xn)), ERR_SHAPE, "resid_layernorm: bad shape");
    if (dtype == BF16) hipLaunchKernelGGL(resid_layernorm_kernel<bf16_t>, dim3(M), dim3(256), 0, st, resid, part, KS, ks_stride, scale, ln_w, ln_b, eps, resid_out, (bf16_t*)xn, H);
    else hipLaunchKernelGGL(resid_layernorm_kernel<float>, dim3(M), dim3(256), 0, st, resid, part, KS, ks_stride, scale, ln_w, ln_b, eps, resid_out, (float*)xn, H);
}

template <typename T>
__global__ __launch_bounds__(256) void rows_to_strips_kernel(const float* __restrict__ src, long S_src, int P_src, const int* __restrict__ lens, int C,
                                                             T* dst, long S_dst, int P_dst) {
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    const float* s = src + ((long)u * S_src + P_src + t) * C;
    T* d = dst + ((long)u * S_dst + P_dst + t) * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) d[c] = from_f32<T>(s[c]);
}
void launch_rows_to_strips(hipStream_t st, int dtype, const float* src, long S_src, int P_src, const int* lens, int U, int Tmax, int C,
                           void* dst, long S_dst, int P_dst) {
    QTTS_CHECK(U >= 1 && Tmax >= 1 && C >= 1, ERR_SHAPE, "rows_to_strips: bad shape");
    dim3 grid(Tmax, U);
    if (dtype == BF16) hipLaunchKernelGGL(rows_to_strips_kernel<bf16_t>, grid, dim3(256), 0, st, src, S_src, P_src, lens, C, (bf16_t*)dst, S_dst, P_dst);
    else hipLaunchKernelGGL(rows_to_strips_kernel<float>, grid, dim3(256), 0, st, src, S_src, P_src, lens, C, (float*)dst, S_dst, P_dst);
}

// EncoderCodebook::encode (quantizer.rs:28-45) + the residual update of EncoderRVQ::encode (:125-130).
template <typename T>
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ dots, const float* __restrict__ c2, const float* __restrict__ emb,
                                                        int bins, int cd, float* residual, T* residual_t, const int* __restrict__ lens, long S,
                                                        unsigned* codes, int T_cap, int Q, int q) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const long m = blockIdx.x;
    const long u = m / S;
    const int t = (int)(m - u * S);
    if (t >= lens[u]) return;
    float best = INFINITY;
    int idx = 0x7fffffff;
    for (int r = threadIdx.x; r < bins; r += 256) {
        const float d = c2[r] - dots[m * bins + r];
        if (d < best || (d == best && r < idx)) { best = d; idx = r; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov < best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    best = bv[0];
    idx = bi[0];
    for (int w = 1; w < 4; ++w)
        if (bv[w] < best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
    if (idx < 0 || idx >= bins) idx = 0;  // every distance was NaN
    if (threadIdx.x == 0) codes[((long)u * T_cap + t) * Q + q] = (unsigned)idx;
    for (int i = threadIdx.x; i < cd; i += 256) {
        const float v = residual[m * cd + i] - emb[(long)idx * cd + i];
        residual[m * cd + i] = v;
        residual_t[m * cd + i] = from_f32<T>(v);
    }
}
void launch_vq_argmin(hipStream_t st, int dtype, const float* dots, const float* c2, const float* emb, int bins, int cd, float* residual,
                      void* residual_t, const int* lens, long S, int U, unsigned* codes, int T_cap, int Q, int q) {
    QTTS_CHECK(bins >= 1 && cd >= 1 && U >= 1 && S >= 1 && q >= 0 && q < Q, ERR_SHAPE, "vq_argmin: bad shape");
    const long M = (long)U * S;
    QTTS_CHECK(M <= 0x7fffffffL, ERR_SHAPE, "vq_argmin: too many rows");
    if (dtype == BF16) hipLaunchKernelGGL(vq_argmin_kernel<bf16_t>, dim3((unsigned)M), dim3(256), 0, st, dots, c2, emb, bins, cd, residual, (bf16_t*)residual_t, lens, S, codes, T_cap, Q, q);
    else hipLaunchKernelGGL(vq_argmin_kernel<float>, dim3((unsigned)M), dim3(256), 0, st, dots, c2, emb, bins, cd, residual, (float*)residual_t, lens, S, codes, T_cap, Q, q);
}

}  // namespace qtts
