// Row-wise / element-wise kernels of the decode path: split-K combine + residual + RMSNorm, SwiGLU,
// bias/activation epilogues, embedding gathers and the per-frame bookkeeping that lets a whole frame
// live inside one hipGraph.  All are HBM/L2-bound byte movers: one 256-thread block per token row,
// float4 accesses, deterministic reduction order.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// ------------------------------------------------------------------------------------------------
// RMSNorm tail shared by the row producers: x (f32, already stored at xrow) -> xn = x / m * w with
// m = sqrt(mean(x^2) + eps)   (qwen_tts/src/nn/norm.rs:17-23 -> candle_nn::rms_norm: (x / m) * w).
template <typename T>
__device__ __forceinline__ void rmsnorm_tail(const float* xrow, float ss_local, const float* __restrict__ w, float eps,
                                             T* xn_row, float* xn_f32_row, int H, float* red, long xn_lo) {
    const float ss = block_sum(ss_local, red);
    const float m = sqrtf(ss / (float)H + eps);
    for (int i = threadIdx.x * 4; i < H; i += blockDim.x * 4) {
        const float4 x = *reinterpret_cast<const float4*>(xrow + i);
        const float4 g = *reinterpret_cast<const float4*>(w + i);
        float4 y = make_float4(x.x / m * g.x, x.y / m * g.y, x.z / m * g.z, x.w / m * g.w);
        if (xn_f32_row) *reinterpret_cast<float4*>(xn_f32_row + i) = y;
        store_split<T>(xn_row + i, xn_lo, y.x);
        store_split<T>(xn_row + i + 1, xn_lo, y.y);
        store_split<T>(xn_row + i + 2, xn_lo, y.z);
        store_split<T>(xn_row + i + 3, xn_lo, y.w);
    }
}

__device__ __forceinline__ float4 resid_combine(const float* __restrict__ resid, const float* __restrict__ part, int KS, long ks_stride,
                                                const float* __restrict__ bias, const float* __restrict__ scale, long off, int i) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part) {
        s = slab_sum4(part + off, KS, ks_stride);
    }
    if (bias) {
        const float4 b = *reinterpret_cast<const float4*>(bias + i);
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
    }
    if (scale) {
        const float4 c = *reinterpret_cast<const float4*>(scale + i);
        s.x *= c.x; s.y *= c.y; s.z *= c.z; s.w *= c.w;
    }
    if (resid) {
        const float4 r = *reinterpret_cast<const float4*>(resid + off);
        s.x = r.x + s.x; s.y = r.y + s.y; s.z = r.z + s.z; s.w = r.w + s.w;
    }
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void resid_norm_kernel(const float* __restrict__ resid, const float* __restrict__ part,
                                                         int KS, long ks_stride, const float* __restrict__ bias,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ norm_w, float eps, float* resid_out,
                                                         T* xn, float* xn_f32, int H, long xn_lo) {
    __shared__ float red[32];
    constexpr int RMAX = 4;  // rows up to 4096 wide stay in registers between the two passes
    const long row = blockIdx.x;
    float ss = 0.f;
    if (H <= RMAX * 1024) {
        // every load of the row (slabs, residual, norm weight) is issued before the first store: resid_out may alias
        // resid, and a store in between would serialise the memory round trips
        float4 keep[RMAX], gw[RMAX];
#pragma unroll
        for (int it = 0; it < RMAX; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            if (i < H) {
                keep[it] = resid_combine(resid, part, KS, ks_stride, bias, scale, row * H + i, i);
                if (norm_w) gw[it] = *reinterpret_cast<const float4*>(norm_w + i);
            }
        }
#pragma unroll
        for (int it = 0; it < RMAX; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            if (i < H) {
                const float4 s = keep[it];
                *reinterpret_cast<float4*>(resid_out + row * H + i) = s;
                ss += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
            }
        }
        if (!norm_w) return;
        const float tot = block_sum(ss, red);
        const float m = sqrtf(tot / (float)H + eps);
#pragma unroll
        for (int it = 0; it < RMAX; ++it) {
            const int i = threadIdx.x * 4 + it * 1024;
            if (i < H) {
                const float4 x = keep[it];
                const float4 g = gw[it];
                const float4 y = make_float4(x.x / m * g.x, x.y / m * g.y, x.z / m * g.z, x.w / m * g.w);
                if (xn_f32) *reinterpret_cast<float4*>(xn_f32 + row * H + i) = y;
                T* o = xn + row * H + i;
                store_split<T>(o, xn_lo, y.x);
                store_split<T>(o + 1, xn_lo, y.y);
                store_split<T>(o + 2, xn_lo, y.z);
                store_split<T>(o + 3, xn_lo, y.w);
            }
        }
        return;
    }
    for (int i = threadIdx.x * 4; i < H; i += blockDim.x * 4) {
        const float4 s = resid_combine(resid, part, KS, ks_stride, bias, scale, row * H + i, i);
        *reinterpret_cast<float4*>(resid_out + row * H + i) = s;
        ss += s.x * s.x + s.y * s.y + s.z * s.z + s.w * s.w;
    }
    if (norm_w) rmsnorm_tail<T>(resid_out + row * H, ss, norm_w, eps, xn + row * H, xn_f32 ? xn_f32 + row * H : nullptr, H, red, xn_lo);
}

void launch_resid_norm(hipStream_t st, int dtype, const float* resid, const float* part, int KS, long ks_stride,
                       const float* bias, const float* scale, const float* norm_w, float eps, float* resid_out, void* xn,
                       float* xn_f32, int M, int H, long xn_lo) {
    QTTS_CHECK(H % 4 == 0 && M >= 1, ERR_SHAPE, "resid_norm: H must be a multiple of 4");
    QTTS_CHECK(resid_out != nullptr && (norm_w == nullptr || xn != nullptr), ERR_BAD_ARG, "resid_norm: null output");
    QTTS_CHECK(xn_lo == 0 || dtype == BF16, ERR_BAD_ARG, "resid_norm: a remainder plane exists for bf16 activations only");
    if (dtype == BF16)
        hipLaunchKernelGGL(resid_norm_kernel<bf16_t>, dim3(M), dim3(256), 0, st, resid, part, KS, ks_stride, bias, scale, norm_w, eps,
                           resid_out, (bf16_t*)xn, xn_f32, H, xn_lo);
    else
        hipLaunchKernelGGL(resid_norm_kernel<float>, dim3(M), dim3(256), 0, st, resid, part, KS, ks_stride, bias, scale, norm_w, eps,
                           resid_out, (float*)xn, xn_f32, H, 0L);
}

// Split-K combine + residual for a consumer that applies the RMSNorm scale itself (the finished-output GEMMs, k_gemm_fused.hip):
// x = resid + sum of slabs + bias;  xb = T(x * norm_w) (+ remainder plane);  ss_out[slice][row] = sum of x^2 over the slice's
// 256 columns.  One wave per (row, slice): no block-wide reduction, eight times the blocks of resid_norm_kernel.
__global__ __launch_bounds__(64) void resid_partial_kernel(const float* __restrict__ resid, const float* __restrict__ part, int KS,
                                                           long ks_stride, const float* __restrict__ bias,
                                                           const float* __restrict__ norm_w, float* resid_out, bf16_t* xb, long xb_lo,
                                                           float* ss_out, long ss_ld, int H) {
    const long row = blockIdx.x;
    const int slice = blockIdx.y;
    const int i = slice * 256 + threadIdx.x * 4;
    const float4 w = *reinterpret_cast<const float4*>(norm_w + i);
    const float4 x = resid_combine(resid, part, KS, ks_stride, bias, nullptr, row * H + i, i);
    *reinterpret_cast<float4*>(resid_out + row * H + i) = x;
    const float y[4] = {x.x * w.x, x.y * w.y, x.z * w.z, x.w * w.w};
    u16 hb[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) hb[q] = f32_to_bf16_bits(y[q]);
    bf16_t* o = xb + row * H + i;
    *reinterpret_cast<uint2*>(o) = make_uint2((unsigned)hb[0] | ((unsigned)hb[1] << 16), (unsigned)hb[2] | ((unsigned)hb[3] << 16));
    if (xb_lo) {
        u16 lb[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) lb[q] = f32_to_bf16_bits(y[q] - bf16_bits_to_f32(hb[q]));
        *reinterpret_cast<uint2*>(o + xb_lo) = make_uint2((unsigned)lb[0] | ((unsigned)lb[1] << 16), (unsigned)lb[2] | ((unsigned)lb[3] << 16));
    }
    const float ss = wave_sum(x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w);
    if (threadIdx.x == 0) ss_out[(long)slice * ss_ld + row] = ss;
}

int launch_resid_partial(hipStream_t st, const float* resid, const float* part, int KS, long ks_stride, const float* bias,
                         const float* norm_w, float* resid_out, void* xb, long xb_lo, float* ss_out, long ss_ld, int M, int H) {
    QTTS_CHECK(H % 256 == 0 && M >= 1 && ss_ld >= M, ERR_SHAPE, "resid_partial: H must be a multiple of 256");
    QTTS_CHECK(resid_out && xb && norm_w && ss_out && xb_lo % 4 == 0, ERR_BAD_ARG, "resid_partial: null output");
    hipLaunchKernelGGL(resid_partial_kernel, dim3(M, H / 256), dim3(64), 0, st, resid, part, KS, ks_stride, bias, norm_w, resid_out,
                       (bf16_t*)xb, xb_lo, ss_out, ss_ld, H);
    return H / 256;
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void swiglu_kernel(const float* __restrict__ part, int KS, long ks_stride, T* act, int I,
                                                     long total4, long act_lo) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total4; e += (long)gridDim.x * blockDim.x) {
        const long row = (e * 4) / I;
        const int i = (int)((e * 4) % I);
        const float* pg = part + row * 2 * I + i;
        const float4 g = slab_sum4(pg, KS, ks_stride);
        const float4 u = slab_sum4(pg + I, KS, ks_stride);
        T* o = act + row * I + i;
        store_split<T>(o, act_lo, g.x / (1.0f + expf(-g.x)) * u.x);
        store_split<T>(o + 1, act_lo, g.y / (1.0f + expf(-g.y)) * u.y);
        store_split<T>(o + 2, act_lo, g.z / (1.0f + expf(-g.z)) * u.z);
        store_split<T>(o + 3, act_lo, g.w / (1.0f + expf(-g.w)) * u.w);
    }
}

void launch_swiglu(hipStream_t st, int dtype, const float* part, int KS, long ks_stride, void* act, int M, int I, long act_lo) {
    QTTS_CHECK(I % 4 == 0 && M >= 1, ERR_SHAPE, "swiglu: I must be a multiple of 4");
    QTTS_CHECK(act_lo == 0 || dtype == BF16, ERR_BAD_ARG, "swiglu: a remainder plane exists for bf16 activations only");
    const long total4 = (long)M * I / 4;
    const int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
    if (dtype == BF16) hipLaunchKernelGGL(swiglu_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, part, KS, ks_stride, (bf16_t*)act, I, total4, act_lo);
    else hipLaunchKernelGGL(swiglu_kernel<float>, dim3(blocks), dim3(256), 0, st, part, KS, ks_stride, (float*)act, I, total4, 0L);
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bias_act_kernel(const float* __restrict__ part, int KS, long ks_stride,
                                                       const float* __restrict__ bias, int act, T* out, int N, long total, long out_lo) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
        float s = part[e];
        for (int k = 1; k < KS; ++k) s += part[k * ks_stride + e];
        if (bias) s += bias[e % N];
        if (act == 1) s = s / (1.0f + expf(-s));
        else if (act == 2) s = 0.5f * s * (1.0f + erff(s * 0.70710678118654752440f));  // exact GELU (Tensor::gelu_erf)
        store_split<T>(out + e, out_lo, s);
    }
}

void launch_bias_act(hipStream_t st, const float* part, int KS, long ks_stride, const float* bias, int act, void* out,
                     int out_dtype, int M, int N, long out_lo) {
    const long total = (long)M * N;
    const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
    QTTS_CHECK(out_lo == 0 || out_dtype == BF16, ERR_BAD_ARG, "bias_act: a remainder plane exists for bf16 activations only");
    if (out_dtype == BF16) hipLaunchKernelGGL(bias_act_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, part, KS, ks_stride, bias, act, (bf16_t*)out, N, total, out_lo);
    else hipLaunchKernelGGL(bias_act_kernel<float>, dim3(blocks), dim3(256), 0, st, part, KS, ks_stride, bias, act, (float*)out, N, total, 0L);
}

// ------------------------------------------------------------------------------------------------
template <typename TS, typename TD>
__global__ __launch_bounds__(256) void gather_rows_kernel(const TS* __restrict__ table, const int* __restrict__ idx, TD* dst,
                                                          int H, long dst_ld) {
    const long row = blockIdx.x;
    const TS* src = table + (long)idx[row] * H;
    for (int i = threadIdx.x; i < H; i += blockDim.x) dst[row * dst_ld + i] = from_f32<TD>(to_f32<TS>(src[i]));
}

void launch_gather_rows(hipStream_t st, const void* table, int tdtype, const int* idx, void* dst, int ddtype, int M, int H,
                        long dst_ld) {
    QTTS_CHECK(M >= 1 && H >= 1, ERR_SHAPE, "gather_rows: empty");
    dim3 g(M), b(256);
    if (tdtype == BF16 && ddtype == BF16) hipLaunchKernelGGL((gather_rows_kernel<bf16_t, bf16_t>), g, b, 0, st, (const bf16_t*)table, idx, (bf16_t*)dst, H, dst_ld);
    else if (tdtype == BF16) hipLaunchKernelGGL((gather_rows_kernel<bf16_t, float>), g, b, 0, st, (const bf16_t*)table, idx, (float*)dst, H, dst_ld);
    else if (ddtype == BF16) hipLaunchKernelGGL((gather_rows_kernel<float, bf16_t>), g, b, 0, st, (const float*)table, idx, (bf16_t*)dst, H, dst_ld);
    else hipLaunchKernelGGL((gather_rows_kernel<float, float>), g, b, 0, st, (const float*)table, idx, (float*)dst, H, dst_ld);
}

template <typename TE>
__global__ __launch_bounds__(256) void assemble_kernel(const float* __restrict__ text_rows, const int* __restrict__ text_idx,
                                                       const TE* __restrict__ codec_emb, const int* __restrict__ codec_id,
                                                       const float* __restrict__ extra, const int* __restrict__ extra_idx,
                                                       float* out, int H) {
    const long row = blockIdx.x;
    const int ti = text_idx[row], ci = codec_id[row], xi = extra_idx[row];
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        // the reference adds the text layer and the codec layer: text + codec (generation.rs:593,723,728,755)
        float v = 0.f;
        bool have = false;
        if (ti >= 0) { v = text_rows[(long)ti * H + i]; have = true; }
        if (ci >= 0) { const float c = to_f32<TE>(codec_emb[(long)ci * H + i]); v = have ? v + c : c; have = true; }
        if (xi >= 0) { const float c = extra[(long)xi * H + i]; v = have ? v + c : c; }
        out[row * H + i] = v;
    }
}

void launch_assemble(hipStream_t st, const float* text_rows, const int* text_idx, const void* codec_emb, int edtype,
                     const int* codec_id, const float* extra, const int* extra_idx, float* out, int M, int H) {
    QTTS_CHECK(M >= 1, ERR_SHAPE, "assemble: empty prompt");
    if (edtype == BF16) hipLaunchKernelGGL(assemble_kernel<bf16_t>, dim3(M), dim3(256), 0, st, text_rows, text_idx, (const bf16_t*)codec_emb, codec_id, extra, extra_idx, out, H);
    else hipLaunchKernelGGL(assemble_kernel<float>, dim3(M), dim3(256), 0, st, text_rows, text_idx, (const float*)codec_emb, codec_id, extra, extra_idx, out, H);
}

// ------------------------------------------------------------------------------------------------
// sum_code_embeddings (qwen_tts/src/nn/talker.rs:461-479) + trailing text / tts_pad
// (nn/generation.rs:1096-1106) + the next layer-0 input norm.  Adds run in codebook order in f32.
template <typename TE, typename T>
__global__ __launch_bounds__(256) void embed_sum_norm_kernel(EmbedSumArgs a, const float* __restrict__ norm_w, float eps,
                                                             float* resid_out, T* xn, long xn_lo) {
    __shared__ float red[32];
    __shared__ const TE* rowp[64];  // embedding row of every codebook, staged so that the gathers below are independent loads
    const int b = blockIdx.x;
    const int step = min(a.step[b], a.max_steps - 1);
    const unsigned* codes = a.codes_base + (long)b * a.row_stride + (long)step * a.G;
    if ((int)threadIdx.x < a.G) {
        const int g = threadIdx.x;
        const TE* table = g == 0 ? (const TE*)a.codec_emb : (const TE*)a.cp_emb[g - 1];
        rowp[g] = table + (long)codes[g] * a.H;
    }
    __syncthreads();
    const float* add = step < a.trailing_len[b] ? a.trailing + ((long)b * a.Tt_max + step) * a.H : a.tts_pad;
    float ss = 0.f;
    // eight consecutive elements per thread: one 16-byte gather per codebook, eight codebooks in flight, adds in codebook order
    for (int i = threadIdx.x * 8; i < a.H; i += blockDim.x * 8) {
        float v[8];
        load8(rowp[0] + i, v);
        int g = 1;
        for (; g + 8 <= a.G; g += 8) {
            float e[8][8];
#pragma unroll
            for (int u = 0; u < 8; ++u) load8(rowp[g + u] + i, e[u]);
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = v[q] + e[u][q];
        }
        for (; g < a.G; ++g) {
            float e[8];
            load8(rowp[g] + i, e);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = v[q] + e[q];
        }
        float t8[8];
        load8(add + i, t8);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            v[q] = v[q] + t8[q];
            ss += v[q] * v[q];
        }
        store8(resid_out + (long)b * a.H + i, v);
    }
    rmsnorm_tail<T>(resid_out + (long)b * a.H, ss, norm_w, eps, xn + (long)b * a.H, nullptr, a.H, red, xn_lo);
}

void launch_embed_sum_norm(hipStream_t st, int dtype, const EmbedSumArgs& a, const float* norm_w, float eps, float* resid_out,
                           void* xn, long xn_lo) {
    QTTS_CHECK(a.B >= 1 && a.H % 8 == 0 && a.G >= 1 && a.G <= 64 && a.max_steps >= 1, ERR_SHAPE, "embed_sum: bad shape");
    if (a.edtype == BF16 && dtype == BF16) hipLaunchKernelGGL((embed_sum_norm_kernel<bf16_t, bf16_t>), dim3(a.B), dim3(256), 0, st, a, norm_w, eps, resid_out, (bf16_t*)xn, xn_lo);
    else if (a.edtype == F32 && dtype == F32) hipLaunchKernelGGL((embed_sum_norm_kernel<float, float>), dim3(a.B), dim3(256), 0, st, a, norm_w, eps, resid_out, (float*)xn, 0L);
    else throw Error(ERR_BAD_ARG, "embed_sum: embedding and activation dtypes must match");
}

// First step of the split RMSNorm (nn/norm.rs:17-23) for rows that do not come out of a GEMM: xb = T(x * w) and the row's sum
// of squares; the consuming GEMM divides by sqrt(ss / H + eps).
template <typename T>
__global__ __launch_bounds__(256) void rows_prep_kernel(const float* __restrict__ x, const float* __restrict__ nw, float* x_out, T* xb, long xb_lo,
                                                        float* ss_out, int H) {
    __shared__ float red[32];
    const long row = blockIdx.x;
    float ss = 0.f;
    for (int i = threadIdx.x; i < H; i += blockDim.x) {
        const float v = x[row * H + i];
        if (x_out) x_out[row * H + i] = v;
        store_split<T>(xb + row * H + i, xb_lo, v * nw[i]);
        ss += v * v;
    }
    const float tot = block_sum(ss, red);
    if (threadIdx.x == 0) ss_out[row] = tot;
}
void launch_rows_prep(hipStream_t st, int dtype, const float* x, const float* nw, float* x_out, void* xb, long xb_lo, float* ss_out, int M, int H) {
    QTTS_CHECK(M >= 1 && H >= 1 && x && nw && xb && ss_out, ERR_BAD_ARG, "rows_prep: bad argument");
    if (dtype == BF16) hipLaunchKernelGGL(rows_prep_kernel<bf16_t>, dim3(M), dim3(256), 0, st, x, nw, x_out, (bf16_t*)xb, xb_lo, ss_out, H);
    else hipLaunchKernelGGL(rows_prep_kernel<float>, dim3(M), dim3(256), 0, st, x, nw, x_out, (float*)xb, 0L, ss_out, H);
}

// sum_ref_code_embeddings (qwen_tts/src/nn/generation.rs:242-273): dst[rows[i]] = codec_emb[c0] + sum_g cp_emb[g-1][c_g]
// for reference frames of an ICL prompt; adds run in codebook order in f32.
template <typename TE>
__global__ __launch_bounds__(256) void sum_code_rows_kernel(const TE* __restrict__ codec_emb, const void* const* cp_emb,
                                                            const int* __restrict__ codes, const int* __restrict__ rows, int G, int H,
                                                            float* dst) {
    const int i = blockIdx.x;
    const int* c = codes + (long)i * G;
    float* o = dst + (long)rows[i] * H;
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        float v = to_f32<TE>(codec_emb[(long)c[0] * H + h]);
        for (int g = 1; g < G; ++g) v = v + to_f32<TE>(((const TE*)cp_emb[g - 1])[(long)c[g] * H + h]);
        o[h] = v;
    }
}
void launch_sum_code_rows(hipStream_t st, int edtype, const void* codec_emb, const void* const* cp_emb, const int* codes, const int* rows,
                          int n, int G, int H, float* dst) {
    QTTS_CHECK(n >= 1 && G >= 1 && H >= 1, ERR_SHAPE, "sum_code_rows: bad shape");
    if (edtype == BF16) hipLaunchKernelGGL(sum_code_rows_kernel<bf16_t>, dim3(n), dim3(256), 0, st, (const bf16_t*)codec_emb, cp_emb, codes, rows, G, H, dst);
    else hipLaunchKernelGGL(sum_code_rows_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)codec_emb, cp_emb, codes, rows, G, H, dst);
}


template <typename T>
__global__ __launch_bounds__(256) void cast_rows_kernel(const float* __restrict__ src, T* dst, long n) {
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) dst[e] = from_f32<T>(src[e]);
}
void launch_cast_rows(hipStream_t st, const float* src, void* dst, int ddtype, long n) {
    const int blocks = (int)std::min<long>((n + 255) / 256, 2048);
    if (ddtype == BF16) hipLaunchKernelGGL(cast_rows_kernel<bf16_t>, dim3(blocks), dim3(256), 0, st, src, (bf16_t*)dst, n);
    else hipLaunchKernelGGL(cast_rows_kernel<float>, dim3(blocks), dim3(256), 0, st, src, (float*)dst, n);
}

// dst[dst_idx[i]][:] = src[src_idx[i]][:]   (same element type; used to pick each row's last prompt position)
template <typename T>
__global__ __launch_bounds__(256) void copy_rows_kernel(const T* __restrict__ src, const int* __restrict__ sidx, T* dst,
                                                        const int* __restrict__ didx, int H) {
    const long s = sidx[blockIdx.x], d = didx[blockIdx.x];
    for (int i = threadIdx.x; i < H; i += blockDim.x) dst[d * H + i] = src[s * H + i];
}
void launch_copy_rows(hipStream_t st, const void* src, const int* sidx, void* dst, const int* didx, int dtype, int n, int H) {
    QTTS_CHECK(n >= 1, ERR_SHAPE, "copy_rows: empty");
    if (dtype == BF16) hipLaunchKernelGGL(copy_rows_kernel<bf16_t>, dim3(n), dim3(256), 0, st, (const bf16_t*)src, sidx, (bf16_t*)dst, didx, H);
    else hipLaunchKernelGGL(copy_rows_kernel<float>, dim3(n), dim3(256), 0, st, (const float*)src, sidx, (float*)dst, didx, H);
}

// ------------------------------------------------------------------------------------------------
__global__ void make_decode_toks_kernel(int* row_len, Tok* toks, int B, int window, const int* active) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const bool live = active == nullptr || active[b] != 0;
    // a retired row recomputes its last position over and over: it never grows past the pages it owns
    const int pos = live ? row_len[b] : max(row_len[b] - 1, 0);
    if (live) row_len[b] = pos + 1;  // the token decoded this step occupies `pos`
    Tok t;
    t.row = b;
    t.pos = pos;
    t.ctx_start = window > 0 ? max(0, pos + 1 - window) : 0;
    t.ctx_len = pos + 1 - t.ctx_start;
    toks[b] = t;
}
void launch_make_decode_toks(hipStream_t st, int* row_len, Tok* toks, int B, int window, const int* active) {
    hipLaunchKernelGGL(make_decode_toks_kernel, dim3((B + 63) / 64), dim3(64), 0, st, row_len, toks, B, window, active);
}

__global__ void advance_step_kernel(int* step, int* active, const int* eos_step, int B, int max_steps, int retire, const int* row_max) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B || active[b] == 0) return;
    const int s = step[b] + 1;
    step[b] = s;
    const int limit = row_max ? min(row_max[b], max_steps) : max_steps;  // a row's own frame limit never exceeds the call's
    if (retire && (s >= limit || (eos_step[b] >= 0 && s > eos_step[b]))) active[b] = 0;
}
void launch_advance_step(hipStream_t st, int* step, int* active, const int* eos_step, int B, int max_steps, int retire, const int* row_max) {
    QTTS_CHECK(step && active && eos_step && B >= 1 && max_steps >= 1, ERR_BAD_ARG, "advance_step: null argument");
    hipLaunchKernelGGL(advance_step_kernel, dim3((B + 63) / 64), dim3(64), 0, st, step, active, eos_step, B, max_steps, retire, row_max);
}

}  // namespace qtts
