// Decode-path GEMM with the consumer's element-wise work fused behind the split-K reduction.
//
// K is never split across blocks (partial sums crossing XCDs inside one launch cost more than they save on this
// part: the per-XCD L2s are not coherent with each other), so every block owns finished output values and applies the
// epilogue itself; no separate residual/RMSNorm, SwiGLU or bias kernel has to be launched:
//   PLAIN   out = [row_scale *] sum + bias                                   (q|k|v projection, logits)
//   RESID   x = resid + sum + bias ; xb = T(x * norm_w) ; ss[tile][row] = sum_cols x^2
//           (qwen_tts/src/nn/decoder_layer.rs:63-86: residual add; the RMSNorm of nn/norm.rs:17-23 is split
//            into the per-column weight applied here and the per-row 1/rms applied by the consuming GEMM)
//   SWIGLU  act = T(silu(g) * u) with gate/up rows interleaved in 16-row groups    (nn/mlp.rs:59-82)
// row_scale[m] = 1 / sqrt(sum_t ss_in[t][m] / H_in + eps) turns the consumer's X = T(x * w) into the RMSNorm'd input.
#include "device_utils.hpp"
#include "kernels.hpp"
#include <algorithm>

namespace qtts {

namespace {

template <typename T>
struct FFrag;
template <>
struct FFrag<bf16_t> {
    static constexpr int CHUNK = 64, EPL = 16;
    bf16x8_t v[2];
    __device__ __forceinline__ void load(const bf16_t* p) {
        v[0] = *reinterpret_cast<const bf16x8_t*>(p);
        v[1] = *reinterpret_cast<const bf16x8_t*>(p + 8);
    }
    static __device__ __forceinline__ f32x4_t mma(const FFrag& a, const FFrag& b, f32x4_t c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[0], b.v[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v[1], b.v[1], c, 0, 0, 0);
        return c;
    }
};
template <>
struct FFrag<float> {
    static constexpr int CHUNK = 32, EPL = 8;
    float v[8];
    __device__ __forceinline__ void load(const float* p) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        const float4 b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ f32x4_t mma(const FFrag& a, const FFrag& b, f32x4_t c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[j], b.v[j], c, 0, 0, 0);
        return c;
    }
};

// One block = one output tile of 16*NT columns for up to 64 rows, full K: the eight waves take contiguous eighths of K
// (register pipeline with up to four chunks in flight, weights and X fragments straight from global memory / L2), their partial tiles are
// summed through LDS in wave order (deterministic), and the whole block applies the epilogue to the finished tile.
constexpr int TILE_WAVES = 8;

template <typename T, int MT, int NT>
__global__ __launch_bounds__(64 * TILE_WAVES) void gemm_tile_kernel(const T* __restrict__ W, const T* __restrict__ X, int N, int K, int ldx,
                                                        GemmEpi e) {
    using F = FFrag<T>;
    constexpr int ROWS = 16 * MT, COLS = 16 * NT;
    constexpr int NW = TILE_WAVES, NTHR = 64 * TILE_WAVES;
    __shared__ f32x4_t red[NW][MT * NT][64];
    __shared__ float rscale[ROWS];
    __shared__ float ssq[ROWS][COLS / 4 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int tile = blockIdx.x;
    const int c0 = tile * COLS;
    const int row0 = blockIdx.z * ROWS;
    const int nC = K / F::CHUNK;
    const int ca = (int)((long)wave * nC / NW), cb = (int)((long)(wave + 1) * nC / NW);

    const T* wp[NT];
    const T* xp[MT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = W + (size_t)(c0 + 16 * j + r) * K + g * F::EPL;
#pragma unroll
    for (int i = 0; i < MT; ++i) xp[i] = X + (size_t)(row0 + 16 * i + r) * ldx + g * F::EPL;
    f32x4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // two register stages of two K chunks each: up to four chunks of weight + X fragments in flight per wave
    F a0[2][NT], b0[2][MT], a1[2][NT], b1[2][MT];
    auto load_stage = [&](F(&a)[2][NT], F(&b)[2][MT], int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (c + u < cb) {
                const size_t off = (size_t)(c + u) * F::CHUNK;
#pragma unroll
                for (int j = 0; j < NT; ++j) a[u][j].load(wp[j] + off);
#pragma unroll
                for (int i = 0; i < MT; ++i) b[u][i].load(xp[i] + off);
            }
        }
    };
    auto mma_stage = [&](const F(&a)[2][NT], const F(&b)[2][MT], int c) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (c + u < cb) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = F::mma(a[u][j], b[u][i], acc[i][j]);
            }
        }
    };
    if (ca < cb) load_stage(a0, b0, ca);
    if (ca + 2 < cb) load_stage(a1, b1, ca + 2);
    // per-row 1/rms of the input, needed only by the epilogue: fetched while the first weight loads are in flight
    if (e.ss_in && (int)threadIdx.x < ROWS) {
        const int m = row0 + threadIdx.x;
        float ss = 0.f;
        for (int t = 0; t < e.ntiles_in; ++t) ss += e.ss_in[(long)t * e.ss_ld + m];
        rscale[threadIdx.x] = sqrtf(ss / (float)e.H_in + e.eps);
    }
    for (int c = ca; c < cb; c += 4) {
        mma_stage(a0, b0, c);
        if (c + 4 < cb) load_stage(a0, b0, c + 4);
        if (c + 2 < cb) mma_stage(a1, b1, c + 2);
        if (c + 6 < cb) load_stage(a1, b1, c + 6);
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) red[wave][i * NT + j][lane] = acc[i][j];
    __syncthreads();

    // ---- epilogue: thread -> (tile t = i * NT + j, lane l): rows 16 i + (l & 15), columns 16 j + 4 (l >> 4) .. + 3
    if (e.mode == GEMM_EPI_SWIGLU) {
        // NT == 2: fragment column j = 0 holds 16 gate rows, j = 1 the matching 16 up rows
        T* act = (T*)e.act;
        const int I = N / 2;
        for (int idx = threadIdx.x; idx < MT * 64; idx += NTHR) {
            const int i = idx >> 6, l = idx & 63;
            const int mi = 16 * i + (l & 15), m = row0 + mi;
            f32x4_t gv = red[0][i * NT][l], uv = red[0][i * NT + (NT - 1)][l];
#pragma unroll
            for (int w = 1; w < NW; ++w) { gv += red[w][i * NT][l]; uv += red[w][i * NT + (NT - 1)][l]; }
            if (e.ss_in) {
                const float d = rscale[mi];
#pragma unroll
                for (int q = 0; q < 4; ++q) { gv[q] = gv[q] / d; uv[q] = uv[q] / d; }
            }
            T* o = act + (size_t)m * I + (c0 / 2) + 4 * (l >> 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(gv[q] / (1.0f + expf(-gv[q])) * uv[q]);
        }
        return;
    }
    for (int idx = threadIdx.x; idx < MT * NT * 64; idx += NTHR) {
        const int t = idx >> 6, l = idx & 63;
        const int i = t / NT, j = t % NT;
        const int mi = 16 * i + (l & 15), m = row0 + mi;
        const int cq = 4 * j + (l >> 4);  // float4 group within the tile row
        const int n = c0 + 4 * cq;
        f32x4_t v = red[0][t][l];
#pragma unroll
        for (int w = 1; w < NW; ++w) v += red[w][t][l];
        if (e.ss_in) {
            const float d = rscale[mi];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = v[q] / d;
        }
        if (e.bias) {
            const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (e.mode == GEMM_EPI_PLAIN) {
            *reinterpret_cast<f32x4_t*>(e.out_f32 + (size_t)m * e.ld_out + n) = v;
        } else {
            if (e.resid) {
                const float4 rr = *reinterpret_cast<const float4*>(e.resid + (size_t)m * N + n);
                v[0] = rr.x + v[0]; v[1] = rr.y + v[1]; v[2] = rr.z + v[2]; v[3] = rr.w + v[3];
            }
            *reinterpret_cast<f32x4_t*>(e.x_out + (size_t)m * N + n) = v;
            float4 w = make_float4(1.f, 1.f, 1.f, 1.f);
            if (e.nw) w = *reinterpret_cast<const float4*>(e.nw + n);
            T* o = (T*)e.xb_out + (size_t)m * N + n;
            o[0] = from_f32<T>(v[0] * w.x);
            o[1] = from_f32<T>(v[1] * w.y);
            o[2] = from_f32<T>(v[2] * w.z);
            o[3] = from_f32<T>(v[3] * w.w);
            ssq[mi][cq] = v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
    }
    if (e.mode == GEMM_EPI_RESID && e.ss_out) {
        __syncthreads();
        if ((int)threadIdx.x < ROWS) {
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < COLS / 4; ++q) s += ssq[threadIdx.x][q];
            e.ss_out[(long)tile * e.ss_ld + row0 + threadIdx.x] = s;
        }
    }
}

struct FusedPlan {
    int MT, NT, Z;
};

FusedPlan plan_fused(int M, int N, int mode) {
    FusedPlan p{};
    p.MT = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    p.Z = (M + 16 * p.MT - 1) / (16 * p.MT);
    // SwiGLU tiles pair 16 gate rows with their 16 up rows; other epilogues use 32 columns once that still leaves
    // at least ~one block per CU, else 16
    if (mode == GEMM_EPI_SWIGLU) p.NT = 2;
    else p.NT = (N % 32 == 0 && (long)(N / 32) * p.Z >= 256) ? 2 : 1;
    return p;
}

template <typename T, int MT>
void launch_tile_mt(hipStream_t st, const FusedPlan& p, const T* W, const T* X, int N, int K, int ldx, const GemmEpi& e) {
    dim3 grid(N / (16 * p.NT), 1, p.Z);
    if (p.NT == 2) hipLaunchKernelGGL((gemm_tile_kernel<T, MT, 2>), grid, dim3(64 * TILE_WAVES), 0, st, W, X, N, K, ldx, e);
    else hipLaunchKernelGGL((gemm_tile_kernel<T, MT, 1>), grid, dim3(64 * TILE_WAVES), 0, st, W, X, N, K, ldx, e);
}

template <typename T>
void launch_tile_t(hipStream_t st, const FusedPlan& p, const T* W, const T* X, int N, int K, int ldx, const GemmEpi& e) {
    if (p.MT == 1) launch_tile_mt<T, 1>(st, p, W, X, N, K, ldx, e);
    else if (p.MT == 2) launch_tile_mt<T, 2>(st, p, W, X, N, K, ldx, e);
    else launch_tile_mt<T, 4>(st, p, W, X, N, K, ldx, e);
}

}  // namespace

int gemm_fused_rows(int M) {
    const int mt = M <= 16 ? 1 : (M <= 32 ? 2 : 4);
    return (M + 16 * mt - 1) / (16 * mt) * 16 * mt;
}
int gemm_fused_ntiles(int M, int N, int mode) {
    const FusedPlan p = plan_fused(M, N, mode);
    return N / (16 * p.NT);
}

void launch_gemm_fused(hipStream_t st, int dtype, const void* W, const void* X, int ldx, int M, int N, int K, const GemmEpi& e) {
    QTTS_CHECK(M >= 1 && N % 32 == 0 && K % 64 == 0, ERR_SHAPE, "gemm_fused: N must be a multiple of 32 and K of 64");
    QTTS_CHECK(e.ss_in == nullptr || (e.ntiles_in >= 1 && e.H_in >= 1), ERR_BAD_ARG, "gemm_fused: bad input-scale description");
    QTTS_CHECK((e.ss_in == nullptr && e.ss_out == nullptr) || e.ss_ld >= gemm_fused_rows(M), ERR_BAD_ARG, "gemm_fused: ss_ld smaller than the padded row count");
    if (e.mode == GEMM_EPI_PLAIN) QTTS_CHECK(e.out_f32 && e.ld_out >= N, ERR_BAD_ARG, "gemm_fused: PLAIN needs an f32 output");
    else if (e.mode == GEMM_EPI_RESID) QTTS_CHECK(e.x_out && e.xb_out, ERR_BAD_ARG, "gemm_fused: RESID needs x / xb outputs");
    else if (e.mode == GEMM_EPI_SWIGLU) QTTS_CHECK(e.act && e.bias == nullptr, ERR_BAD_ARG, "gemm_fused: SWIGLU needs an activation output and no bias");
    else throw Error(ERR_BAD_ARG, "gemm_fused: unknown epilogue");
    const FusedPlan p = plan_fused(M, N, e.mode);
    if (dtype == BF16) launch_tile_t<bf16_t>(st, p, (const bf16_t*)W, (const bf16_t*)X, N, K, ldx, e);
    else launch_tile_t<float>(st, p, (const float*)W, (const float*)X, N, K, ldx, e);
}

}  // namespace qtts
