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

// One float4 of GEMM output: row m of the GEMM row space (strip u = m / S_in, step tp = m % S_in), columns n .. n + 3.
template <typename T>
__device__ __forceinline__ void conv_epi_apply(const ConvEpi& e, long u, int tp, int n, float (&v)[4]) {
    const int j = n / e.Cout, co = n % e.Cout;
    const long orow = u * e.S_out + e.P_out + (long)tp * e.r + j;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float x = v[q] + (e.bias ? e.bias[co + q] : 0.f);
        if (e.bias_u) x += e.bias_u[u * e.Cout + co + q];
        if (e.act == 1) x = gelu_tanh(x);
        else if (e.act >= 2) {
            x = fmaxf(x, 0.f);
            if (e.act == 3) x = tanhf(x);
        }
        if (e.scale) x = x * e.scale[co + q];
        if (e.resid) x = to_f32<T>(((const T*)e.resid)[orow * e.ld_out + co + q]) + x;
        if (e.resid_f32) x = e.resid_f32[orow * e.ld_out + co + q] + x;
        v[q] = x;
    }
    if (e.Cvalid && co >= e.Cvalid) return;  // columns past the layer's real width are padding
    if (e.out) {
        T* o = (T*)e.out + orow * e.ld_out + co;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(v[q]);
    }
    if (e.out_f32) *reinterpret_cast<float4*>(e.out_f32 + orow * e.ld_out + co) = make_float4(v[0], v[1], v[2], v[3]);
    if (e.out2) {
        T* o = (T*)e.out2 + orow * e.ld_out + co;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // SnakeBeta in f32 on the value the next op would read: the stored (rounded) one unless an f32 copy is kept
            const float x = e.out_f32 ? v[q] : to_f32<T>(from_f32<T>(v[q]));
            const float sn = sinf(x * e.alpha_exp[co + q]);
            o[q] = from_f32<T>(x + sn * sn / e.beta_den[co + q]);
        }
    }
}

}  // namespace qtts
