// Mel front end and the element-wise / pooling kernels of the ECAPA-TDNN speaker encoder
// (qwen_tts/src/audio/mel.rs:81-230, qwen_tts/src/nn/speaker_encoder.rs:20-467).
//
// Activations are channels-last strips, one per utterance: strip u occupies rows [u*S, (u+1)*S); time step t of
// utterance u is row u*S + P + t.  Convolutions with kernel > 1 use reflect "same" padding, materialised as
// reflected copies in the rows next to the valid range, so that every conv is a tap-addressed GEMM (k_gemm.hip).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// ------------------------------------------------------------------------------------------------
// One STFT frame per block: reflect-pad 384, periodic Hann, 1024-point radix-2 FFT in LDS, magnitude
// sqrt(re^2 + im^2 + 1e-9), 128 Slaney mel filters, ln(max(., 1e-5)).
constexpr int NFFT = 1024, HOP = 256, NMEL = 128, NBIN = NFFT / 2 + 1, MELPAD = (NFFT - HOP) / 2;

template <typename T>
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ audio, const int* __restrict__ n_samples, int max_samples,
                                                  const float* __restrict__ window, const float2* __restrict__ twiddle /*[512]*/,
                                                  const float* __restrict__ fbT /*[513][128]*/, float* mel_out, long frames_stride,
                                                  T* strips, long S, int P) {
    __shared__ float2 buf[NFFT];
    __shared__ float mag[NBIN + 3];
    const int u = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
    const int n = n_samples[u];
    const int frames = (n + 2 * MELPAD - NFFT) / HOP + 1;
    if (f >= frames) return;
    const float* x = audio + (long)u * max_samples;
    for (int i = tid; i < NFFT; i += 256) {
        int s = f * HOP + i - MELPAD;
        if (s < 0) s = -s;                      // reflect without repeating the edge sample (mel.rs:161-181)
        else if (s >= n) s = 2 * (n - 1) - s;
        const float v = x[s] * window[i];
        buf[__brev((unsigned)i) >> 22] = make_float2(v, 0.f);
    }
    __syncthreads();
    for (int len = 2; len <= NFFT; len <<= 1) {
        const int half = len >> 1, step = NFFT / len;
        for (int b = tid; b < NFFT / 2; b += 256) {
            const int k = b % half, base = (b / half) * len;
            const float2 w = twiddle[k * step];
            const float2 a = buf[base + k], c = buf[base + k + half];
            const float2 v = make_float2(c.x * w.x - c.y * w.y, c.x * w.y + c.y * w.x);
            buf[base + k] = make_float2(a.x + v.x, a.y + v.y);
            buf[base + k + half] = make_float2(a.x - v.x, a.y - v.y);
        }
        __syncthreads();
    }
    for (int b = tid; b < NBIN; b += 256) {
        const float2 c = buf[b];
        mag[b] = sqrtf(c.x * c.x + c.y * c.y + 1e-9f);
    }
    __syncthreads();
    // two threads per mel bin (even / odd frequency bins), combined with one shuffle
    const int mbin = tid >> 1, par = tid & 1;
    float s = 0.f;
    for (int b = par; b < NBIN; b += 2) s += fbT[b * NMEL + mbin] * mag[b];
    s += __shfl_xor(s, 1, 64);
    if (par == 0) {
        const float v = logf(fmaxf(s, 1e-5f));
        if (mel_out) mel_out[((long)u * frames_stride + f) * NMEL + mbin] = v;
        if (strips) strips[((long)u * S + P + f) * NMEL + mbin] = from_f32<T>(v);
    }
}

void launch_mel(hipStream_t st, int dtype, const float* audio, const int* n_samples, int B, int max_samples, int max_frames,
                const float* window, const float* twiddle, const float* fbT, float* mel_out, long frames_stride, void* strips, long S,
                int P) {
    QTTS_CHECK(B >= 1 && max_frames >= 1 && max_samples > MELPAD, ERR_SHAPE, "mel: bad shape");
    dim3 grid(max_frames, B);
    if (dtype == BF16)
        hipLaunchKernelGGL(mel_kernel<bf16_t>, grid, dim3(256), 0, st, audio, n_samples, max_samples, window, (const float2*)twiddle, fbT,
                           mel_out, frames_stride, (bf16_t*)strips, S, P);
    else
        hipLaunchKernelGGL(mel_kernel<float>, grid, dim3(256), 0, st, audio, n_samples, max_samples, window, (const float2*)twiddle, fbT,
                           mel_out, frames_stride, (float*)strips, S, P);
}

// ------------------------------------------------------------------------------------------------
// reflect_pad_1d (speaker_encoder.rs:64-108): rows for t in [-pad, 0) and [T, T + pad) copy t' = -t / 2(T-1) - t.
template <typename T>
__global__ __launch_bounds__(256) void reflect_fill_kernel(T* x, const int* __restrict__ lens, long S, int P, int pad, int C) {
    const int u = blockIdx.y, i = blockIdx.x % pad, side = blockIdx.x / pad;
    const int Tu = lens[u];
    if (i + 1 >= Tu) return;  // caller guarantees pad < T; stay in bounds regardless
    const long dst = (long)u * S + P + (side == 0 ? -(i + 1) : Tu + i);
    const long src = (long)u * S + P + (side == 0 ? (i + 1) : Tu - 2 - i);
    for (int c = threadIdx.x; c < C; c += blockDim.x) x[dst * C + c] = x[src * C + c];
}
void launch_reflect_fill(hipStream_t st, int dtype, void* x, const int* lens, int U, long S, int P, int pad, int C) {
    QTTS_CHECK(pad >= 1 && pad <= P && U >= 1, ERR_SHAPE, "reflect_fill: bad padding");
    dim3 grid(2 * pad, U);
    if (dtype == BF16) hipLaunchKernelGGL(reflect_fill_kernel<bf16_t>, grid, dim3(256), 0, st, (bf16_t*)x, lens, S, P, pad, C);
    else hipLaunchKernelGGL(reflect_fill_kernel<float>, grid, dim3(256), 0, st, (float*)x, lens, S, P, pad, C);
}

// Res2Net step input (speaker_encoder.rs:156-183): tmp = a[:, j*cs:(j+1)*cs] (+ previous chunk's output), written
// with its reflected borders into a 64-channel strip (channels >= cs are zero).
template <typename T>
__global__ __launch_bounds__(64) void res2net_prep_kernel(const T* __restrict__ a, const T* __restrict__ prev, const int* __restrict__ lens,
                                                          long S, int P, int pad, int ld, int off, int cs, T* tmp) {
    const int u = blockIdx.y, Tu = lens[u];
    const int tt = (int)blockIdx.x - pad;  // time in [-pad, Tmax + pad)
    if (tt >= Tu + pad) return;
    int ts = tt < 0 ? -tt : (tt >= Tu ? 2 * (Tu - 1) - tt : tt);
    if (ts < 0 || ts >= Tu) return;
    const long src = (long)u * S + P + ts, dst = (long)u * S + P + tt;
    const int c = threadIdx.x;
    float v = 0.f;
    if (c < cs) {
        v = to_f32<T>(a[src * ld + off + c]);
        if (prev) v = to_f32<T>(from_f32<T>(v + to_f32<T>(prev[src * ld + off - cs + c])));
    }
    tmp[dst * 64 + c] = from_f32<T>(v);
}
void launch_res2net_prep(hipStream_t st, int dtype, const void* a, const void* prev, const int* lens, int U, int Tmax, long S, int P,
                         int pad, int ld, int off, int cs, void* tmp) {
    QTTS_CHECK(cs >= 1 && cs <= 64 && pad <= P && off + cs <= ld && (prev == nullptr || off >= cs), ERR_SHAPE, "res2net_prep: bad shape");
    dim3 grid(Tmax + 2 * pad, U);
    if (dtype == BF16) hipLaunchKernelGGL(res2net_prep_kernel<bf16_t>, grid, dim3(64), 0, st, (const bf16_t*)a, (const bf16_t*)prev, lens, S, P, pad, ld, off, cs, (bf16_t*)tmp);
    else hipLaunchKernelGGL(res2net_prep_kernel<float>, grid, dim3(64), 0, st, (const float*)a, (const float*)prev, lens, S, P, pad, ld, off, cs, (float*)tmp);
}

// copy a channel slice (Res2Net chunk 0 passes through)
template <typename T>
__global__ __launch_bounds__(64) void slice_copy_kernel(const T* __restrict__ a, const int* __restrict__ lens, long S, int P, int ld, int cs,
                                                        T* out) {
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    const long row = (long)u * S + P + t;
    if ((int)threadIdx.x < cs) out[row * ld + threadIdx.x] = a[row * ld + threadIdx.x];
}
void launch_slice_copy(hipStream_t st, int dtype, const void* a, const int* lens, int U, int Tmax, long S, int P, int ld, int cs, void* out) {
    QTTS_CHECK(cs >= 1 && cs <= 64, ERR_SHAPE, "slice_copy: bad shape");
    dim3 grid(Tmax, U);
    if (dtype == BF16) hipLaunchKernelGGL(slice_copy_kernel<bf16_t>, grid, dim3(64), 0, st, (const bf16_t*)a, lens, S, P, ld, cs, (bf16_t*)out);
    else hipLaunchKernelGGL(slice_copy_kernel<float>, grid, dim3(64), 0, st, (const float*)a, lens, S, P, ld, cs, (float*)out);
}

// ------------------------------------------------------------------------------------------------
// Per-(utterance, channel) statistics over time.  One thread per channel, sequential in t (the summation order
// of the reference's mean / weighted sums), coalesced across channels.
//   mode 0: mean over t                                       -> out[u][c]              (SE squeeze, :218-222)
//   mode 1: uniform-weight mean and std (eps 1e-12)           -> out[u][c], out[u][C+c]  (ASP global context, :398-430)
//   mode 2: softmax over t of (logits + bias) then weighted mean / std -> out[u][c], out[u][C+c]  (:440-456)
template <typename T>
__global__ __launch_bounds__(256) void time_stats_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ logits, int ldl,
                                                         const float* __restrict__ lbias, const int* __restrict__ lens, long S, int P,
                                                         int Pl, int C, int mode, float* out) {
    const int u = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int Tu = lens[u];
    const long row0 = (long)u * S + P;
    const T* xp = x + row0 * ldx + c;
    if (mode == 0) {
        float s = 0.f;
        for (int t = 0; t < Tu; ++t) s += to_f32<T>(xp[(long)t * ldx]);
        out[(long)u * C + c] = s / (float)Tu;
        return;
    }
    float mu = 0.f, var = 0.f;
    if (mode == 1) {
        const float w = 1.0f / (float)Tu;
        for (int t = 0; t < Tu; ++t) mu += to_f32<T>(xp[(long)t * ldx]) * w;
        for (int t = 0; t < Tu; ++t) {
            const float d = to_f32<T>(xp[(long)t * ldx]) - mu;
            var += d * d * w;
        }
    } else {
        const float* lp = logits + ((long)u * S + Pl) * ldl + c;
        const float b = lbias ? lbias[c] : 0.f;
        float mx = -INFINITY;
        for (int t = 0; t < Tu; ++t) mx = fmaxf(mx, lp[(long)t * ldl] + b);
        float den = 0.f;
        for (int t = 0; t < Tu; ++t) den += expf(lp[(long)t * ldl] + b - mx);
        for (int t = 0; t < Tu; ++t) mu += to_f32<T>(xp[(long)t * ldx]) * (expf(lp[(long)t * ldl] + b - mx) / den);
        for (int t = 0; t < Tu; ++t) {
            const float d = to_f32<T>(xp[(long)t * ldx]) - mu;
            var += d * d * (expf(lp[(long)t * ldl] + b - mx) / den);
        }
    }
    out[(long)u * 2 * C + c] = mu;
    out[(long)u * 2 * C + C + c] = sqrtf(var + 1e-12f);
}
void launch_time_stats(hipStream_t st, int dtype, const void* x, int ldx, const float* logits, int ldl, const float* lbias, const int* lens,
                       int U, long S, int P, int Pl, int C, int mode, float* out) {
    QTTS_CHECK(U >= 1 && C >= 1 && C <= ldx && mode >= 0 && mode <= 2 && (mode != 2 || (logits && C <= ldl)), ERR_SHAPE, "time_stats: bad shape");
    dim3 grid((C + 255) / 256, U);
    if (dtype == BF16) hipLaunchKernelGGL(time_stats_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, logits, ldl, lbias, lens, S, P, Pl, C, mode, out);
    else hipLaunchKernelGGL(time_stats_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, logits, ldl, lbias, lens, S, P, Pl, C, mode, out);
}

// y[u][n] = act(b[n] + sum_k W[n][k] x[u][k]) for small per-utterance vectors (SE gates, ASP context term, final fc).
// One wave per output; act 0 none, 2 ReLU, 4 sigmoid.
__global__ __launch_bounds__(256) void small_fc_kernel(const float* __restrict__ W, int ldw, const float* __restrict__ b,
                                                       const float* __restrict__ x, int ldx, int N, int K, int act, float* y, int ldy) {
    const int u = blockIdx.y, n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += W[(long)n * ldw + k] * x[(long)u * ldx + k];
    s = wave_sum(s);
    if (lane == 0) {
        s += b ? b[n] : 0.f;
        if (act == 2) s = fmaxf(s, 0.f);
        else if (act == 4) s = 1.0f / (1.0f + expf(-s));
        y[(long)u * ldy + n] = s;
    }
}
void launch_small_fc(hipStream_t st, const float* W, int ldw, const float* b, const float* x, int ldx, int U, int N, int K, int act, float* y,
                     int ldy) {
    QTTS_CHECK(U >= 1 && N >= 1 && K >= 1 && K <= ldw && K <= ldx && N <= ldy, ERR_SHAPE, "small_fc: bad shape");
    hipLaunchKernelGGL(small_fc_kernel, dim3((N + 3) / 4, U), dim3(256), 0, st, W, ldw, b, x, ldx, N, K, act, y, ldy);
}

// SE excitation + residual (speaker_encoder.rs:225-230, 299-315): out = [resid +] x * gate[u]
template <typename T>
__global__ __launch_bounds__(256) void se_apply_kernel(const T* __restrict__ x, int ldx, const float* __restrict__ gate, const T* resid,
                                                       int ldr, const int* __restrict__ lens, long S, int P, int C, T* out, int ldo) {
    const int u = blockIdx.y, t = blockIdx.x;
    if (t >= lens[u]) return;
    const long row = (long)u * S + P + t;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float v = to_f32<T>(from_f32<T>(to_f32<T>(x[row * ldx + c]) * gate[(long)u * C + c]));
        if (resid) v = to_f32<T>(resid[row * ldr + c]) + v;
        out[row * ldo + c] = from_f32<T>(v);
    }
}
void launch_se_apply(hipStream_t st, int dtype, const void* x, int ldx, const float* gate, const void* resid, int ldr, const int* lens, int U,
                     int Tmax, long S, int P, int C, void* out, int ldo) {
    QTTS_CHECK(U >= 1 && C <= ldx && C <= ldo && (resid == nullptr || C <= ldr), ERR_SHAPE, "se_apply: bad shape");
    dim3 grid(Tmax, U);
    if (dtype == BF16) hipLaunchKernelGGL(se_apply_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ldx, gate, (const bf16_t*)resid, ldr, lens, S, P, C, (bf16_t*)out, ldo);
    else hipLaunchKernelGGL(se_apply_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ldx, gate, (const float*)resid, ldr, lens, S, P, C, (float*)out, ldo);
}

}  // namespace qtts
