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
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

template <typename T>
struct Frag;
template <>
struct Frag<bf16_t> {
    static constexpr int CHUNK = 64;  // K elements covered by one wave-wide 32-byte-per-lane load
    static constexpr int EPL = 16;    // elements per lane
    bf16x8_t v[2];
    __device__ __forceinline__ void load(const bf16_t* p) {
        v[0] = *reinterpret_cast<const bf16x8_t*>(p);
        v[1] = *reinterpret_cast<const bf16x8_t*>(p + 8);
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
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ f32x4_t mma(const Frag& a, const Frag& b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
        return c;
    }
};

template <typename T, int MT, int NT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(const T* __restrict__ W, const T* __restrict__ X,
                                                          float* __restrict__ out, int N, int K, int KS, int ldx,
                                                          long out_ks_stride) {
    using F = Frag<T>;
    __shared__ f32x4_t red[4][MT * NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (16 * NT);
    const int ks = blockIdx.y;
    const int row0 = blockIdx.z * (16 * MT);
    const int nC = K / F::CHUNK, S = KS * 4, s = ks * 4 + wave;
    const int c0 = (int)((long)s * nC / S), c1 = (int)((long)(s + 1) * nC / S);

    const T* wp[NT];
    const T* xp[MT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = W + (size_t)(n0 + 16 * j + r) * K + g * F::EPL;
#pragma unroll
    for (int i = 0; i < MT; ++i) xp[i] = X + (size_t)(row0 + 16 * i + r) * ldx + g * F::EPL;

    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};


    for (int c = c0; c < c1; ++c) {
        F a[NT], b[MT];
        const size_t off = (size_t)c * F::CHUNK;
#pragma unroll
        for (int j = 0; j < NT; ++j) a[j].load(wp[j] + off);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[i].load(xp[i] + off);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[j], b[i], acc[i][j]);
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

template <typename T, int MT>
static void launch_mt(hipStream_t st, const T* W, const T* X, float* out, int Mpad, int N, int K, int KS, int ldx) {
    const int Z = Mpad / (16 * MT);
    const long kstride = (long)Mpad * N;
    if (N % 32 == 0 && (long)(N / 32) * KS * Z >= 256) {
        dim3 grid(N / 32, KS, Z);
        hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 2>), grid, dim3(256), 0, st, W, X, out, N, K, KS, ldx, kstride);
    } else {
        dim3 grid(N / 16, KS, Z);
        hipLaunchKernelGGL((gemm_skinny_kernel<T, MT, 1>), grid, dim3(256), 0, st, W, X, out, N, K, KS, ldx, kstride);
    }
}

template <typename T>
static void launch_t(hipStream_t st, const T* W, const T* X, float* out, int Mpad, int N, int K, int KS, int ldx) {
    if (Mpad == 16) launch_mt<T, 1>(st, W, X, out, Mpad, N, K, KS, ldx);
    else if (Mpad == 32) launch_mt<T, 2>(st, W, X, out, Mpad, N, K, KS, ldx);
    else launch_mt<T, 4>(st, W, X, out, Mpad, N, K, KS, ldx);
}

int gemm_pad_rows(int M) { return M <= 16 ? 16 : (M <= 32 ? 32 : (M + 63) / 64 * 64); }

int gemm_pick_ks(int M, int N, int K, int dtype) {
    const int chunk = dtype == BF16 ? 64 : 32;
    const int nC = K / chunk;
    const int Mpad = gemm_pad_rows(M);
    const int Z = Mpad <= 32 ? 1 : Mpad / 64;
    const long nblk = (long)(N / 16) * Z;  // blocks at KS = 1 with one 16-row tile each
    int ks = 1;
    // aim for >= ~512 blocks (2 per CU) while every wave keeps at least two chunks
    while (nblk * ks < 512 && nC / (4 * ks * 2) >= 2 && ks < 16) ks *= 2;
    return ks;
}

void launch_gemm(hipStream_t st, int dtype, const void* W, const void* X, float* out, int M, int N, int K, int KS, int ldx) {
    QTTS_CHECK(N % 16 == 0, ERR_SHAPE, "gemm: N must be a multiple of 16");
    QTTS_CHECK(K % 64 == 0, ERR_SHAPE, "gemm: K must be a multiple of 64");
    QTTS_CHECK(KS >= 1 && M >= 1, ERR_SHAPE, "gemm: bad M/KS");
    const int Mpad = gemm_pad_rows(M);
    if (dtype == BF16) launch_t<bf16_t>(st, (const bf16_t*)W, (const bf16_t*)X, out, Mpad, N, K, KS, ldx);
    else launch_t<float>(st, (const float*)W, (const float*)X, out, Mpad, N, K, KS, ldx);
}

}  // namespace qtts
