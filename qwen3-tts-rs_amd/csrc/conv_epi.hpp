// Element-wise tail of every conv / linear GEMM of the codec decoder and the speaker encoder, shared by the
// standalone epilogue kernel (k_codec.hip) and the large-M GEMM that applies it in registers (k_gemm.hip):
// bias, optional tanh-GELU (audio/tokenizer/v2/convnext.rs:81) / ReLU / tanh(ReLU), per-channel scale (ConvNeXt gamma),
// residual, then the raw result and/or its SnakeBeta (audio/tokenizer/v2/snake_beta.rs:37-73:
// x + sin^2(x e^alpha) / (e^beta + 1e-9), in f32).
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

__device__ __forceinline__ float gelu_tanh(float v) {
    return 0.5f * v * (1.0f + tanhf(0.7978845608028654f * v * (1.0f + 0.044715f * v * v)));
}

// SnakeBeta, x + sin^2(x e^alpha) / (e^beta + 1e-9), from the per-channel pair (al, dn) = (e^alpha, e^beta + 1e-9).
// f32 runs keep the reference's arithmetic (full-precision sine, a division: bit-closeness to its libm).  bf16 runs evaluate
// sin^2 t = (1 - cos 2t) / 2 with the hardware cosine on an argument reduced to [0, 1) revolutions: snake_prep turns the pair
// into (e^alpha / pi, 0.5 / dn) once per channel, and an element then costs five instructions, one of them transcendental
// (multiply, v_fract, v_cos, add, fma) instead of eight with two; the ~1e-6 absolute error is far below bf16 rounding.
template <typename T>
__device__ __forceinline__ void snake_prep(float& al, float& dn);
template <>
__device__ __forceinline__ void snake_prep<float>(float&, float&) {}
template <>
__device__ __forceinline__ void snake_prep<bf16_t>(float& al, float& dn) {
    al *= 0.3183098861837907f;
    dn = 0.5f * __builtin_amdgcn_rcpf(dn);
}
template <typename T>
__device__ __forceinline__ float snake_apply(float x, float al, float dn);
template <>
__device__ __forceinline__ float snake_apply<float>(float x, float al, float dn) {
    const float sn = sinf(x * al);
    return x + sn * sn / dn;
}
template <>
__device__ __forceinline__ float snake_apply<bf16_t>(float x, float al, float dn) {
    const float c = __builtin_amdgcn_cosf(__builtin_amdgcn_fractf(x * al));
    return fmaf(-dn, c, x + dn);
}

// four consecutive elements of a model-dtype tensor (aligned to four elements) as f32
template <typename T>
__device__ __forceinline__ void load4(const T* p, float (&o)[4]);
template <>
__device__ __forceinline__ void load4<float>(const float* p, float (&o)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&o)[4]) {
    const uint2 t = *reinterpret_cast<const uint2*>(p);
    o[0] = __uint_as_float(t.x << 16); o[1] = __uint_as_float(t.x & 0xffff0000u);
    o[2] = __uint_as_float(t.y << 16); o[3] = __uint_as_float(t.y & 0xffff0000u);
}

__device__ __forceinline__ void ld4(const float* p, float (&o)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
}

// One float4 of GEMM output: row m of the GEMM row space (strip u = m / S_in, step tp = m % S_in), columns n .. n + 3
// (n is a multiple of 4, so are Cout and every channel offset: all parameter vectors are read as float4).
// The tail is split in three so that a kernel can hoist what depends on the columns only (conv_epi_cols), issue the
// residual loads of several rows before it needs any of them (conv_epi_resid), and then do the arithmetic and the
// stores (conv_epi_finish); conv_epi_apply is the three in a row.
struct EpiCols {
    int j, co;
    float bias[4], scale[4], al[4], den[4];  // al / den: SnakeBeta parameters as snake_prep leaves them
};

template <typename T>
__device__ __forceinline__ void conv_epi_cols(const ConvEpi& e, int n, EpiCols& c) {
    c.j = 0;
    c.co = n;
    if (e.r != 1) { c.j = n / e.Cout; c.co = n - c.j * e.Cout; }
#pragma unroll
    for (int q = 0; q < 4; ++q) { c.bias[q] = 0.f; c.scale[q] = 1.f; c.al[q] = 0.f; c.den[q] = 1.f; }
    if (e.bias) ld4(e.bias + c.co, c.bias);
    if (e.scale) ld4(e.scale + c.co, c.scale);
    if (e.out2 && e.act2 != 1) {
        ld4(e.alpha_exp + c.co, c.al);
        ld4(e.beta_den + c.co, c.den);
#pragma unroll
        for (int q = 0; q < 4; ++q) snake_prep<T>(c.al[q], c.den[q]);
    }
}

__device__ __forceinline__ long conv_epi_offset(const ConvEpi& e, const EpiCols& c, long u, int tp) {
    const long orow = u * e.S_out + e.P_out + (long)tp * e.r + c.j;
    return orow * e.ld_out + c.co;
}

template <typename T>
__device__ __forceinline__ void conv_epi_resid(const ConvEpi& e, long o_off, float (&rr)[4]) {
    if (e.resid) {
        const T* rp = (const T*)e.resid + o_off;
        load4<T>(rp, rr);
        if (e.resid_lo) {
            float lo[4];
            load4<T>(rp + e.resid_lo, lo);
#pragma unroll
            for (int q = 0; q < 4; ++q) rr[q] += lo[q];
        }
    } else if (e.resid_f32) {
        ld4(e.resid_f32 + o_off, rr);
    }
}

template <typename T>
__device__ __forceinline__ void conv_epi_finish(const ConvEpi& e, const EpiCols& c, long u, long o_off, float (&v)[4], const float (&rr)[4]) {
    if (e.bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += c.bias[q];
    }
    if (e.bias_u) {
        float b[4];
        ld4(e.bias_u + u * e.Cout + c.co, b);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += b[q];
    }
    if (e.act == 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = gelu_tanh(v[q]);
    } else if (e.act >= 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (e.act == 4) v[q] = v[q] > 0.f ? v[q] : expf(v[q]) - 1.0f;  // ELU(alpha = 1)
            else {
                v[q] = fmaxf(v[q], 0.f);
                if (e.act == 3) v[q] = tanhf(v[q]);
            }
        }
    }
    if (e.scale) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = v[q] * c.scale[q];
    }
    if (e.resid || e.resid_f32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = rr[q] + v[q];
    }
    if (e.Cvalid && c.co >= e.Cvalid) return;  // columns past the layer's real width are padding
    if (e.out) store4_split<T>((T*)e.out + o_off, e.out_lo, v);
    if (e.out_f32) *reinterpret_cast<float4*>(e.out_f32 + o_off) = make_float4(v[0], v[1], v[2], v[3]);
    if (e.out2 && e.act2 == 1) {  // ELU of the value the next op would read
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float x = (e.out_f32 || e.resid_f32 || e.out_lo || e.out2_lo) ? v[q] : to_f32<T>(from_f32<T>(v[q]));
            y[q] = x > 0.f ? x : expf(x) - 1.0f;
        }
        store4_split<T>((T*)e.out2 + o_off, e.out2_lo, y);
    } else if (e.out2) {
        float y[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // SnakeBeta in f32 on the value the next op would read: the stored (rounded) one unless the stream is kept in f32
            // or with its remainder
            const float x = (e.out_f32 || e.resid_f32 || e.out_lo || e.out2_lo) ? v[q] : to_f32<T>(from_f32<T>(v[q]));
            y[q] = snake_apply<T>(x, c.al[q], c.den[q]);
        }
        store4_split<T>((T*)e.out2 + o_off, e.out2_lo, y);
    }
}

template <typename T>
__device__ __forceinline__ void conv_epi_apply(const ConvEpi& e, long u, int tp, int n, float (&v)[4]) {
    EpiCols c;
    conv_epi_cols<T>(e, n, c);
    const long o_off = conv_epi_offset(e, c, u, tp);
    float rr[4] = {0.f, 0.f, 0.f, 0.f};
    conv_epi_resid<T>(e, o_off, rr);
    conv_epi_finish<T>(e, c, u, o_off, v, rr);
}

}  // namespace qtts
