// On-device token sampling: the reference's logit processors in their HF order and its inverse-CDF
// draw (qwen_tts/src/nn/sampling.rs:310-397), one 256-thread block per batch row, logits staged in LDS.
//   repetition penalty (applied once per earlier occurrence, :184-193) -> suppress range / EOS while
//   history < min_new_tokens (:319-339) -> * 1/T (:199-204) -> top-k: keep >= k-th largest (:209-229)
//   -> top-p: stable sort by prob, shortest prefix with cumsum > p (:234-274) -> argmax, or softmax +
//   renormalised running sum against one 24-bit uniform (:356-397).
// The sequential f32 sums of the reference are reproduced by a single lane walking the (few) surviving
// candidates in index order; adding the exact zeros of removed candidates would not change them.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

constexpr int SAMPLE_MAX_V = 4096;

__device__ __forceinline__ unsigned okey(float f) {  // order-preserving map float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

__global__ __launch_bounds__(256) void sample_kernel(SampleArgs a) {
    __shared__ float lg[SAMPLE_MAX_V];
    __shared__ float sv[SAMPLE_MAX_V];          // survivor values (logit, later prob)
    __shared__ unsigned short si[SAMPLE_MAX_V]; // survivor vocabulary index
    __shared__ unsigned short order[SAMPLE_MAX_V];
    __shared__ int hist[256];
    __shared__ int cnt[257];
    __shared__ float redf[8];
    __shared__ int redi[8];
    __shared__ int sh_sel, sh_k, sh_n, sh_tok;
    __shared__ float sh_f;

    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    // a retired row (active[b] == 0) still runs - the launch covers every slot - on its last frame index, and writes nothing
    const bool live = a.active == nullptr || a.active[b] != 0;
    const int step = min(a.step[b], a.max_steps - 1);
    const float* src = a.logits + (long)b * V;
    const int* hc = a.hist_count ? a.hist_count + (long)b * V : nullptr;
    const bool eos_blocked = a.eos_id >= 0 && step < a.min_new_tokens;

    for (int i = tid; i < V; i += 256) {
        float v = src[i];
        if (hc && a.rep_penalty != 1.0f) {
            const int c = hc[i];
            for (int r = 0; r < c; ++r) v = v > 0.0f ? v / a.rep_penalty : v * a.rep_penalty;
        }
        if ((i >= a.suppress_lo && i < a.suppress_hi && i != a.eos_id) || (eos_blocked && i == a.eos_id)) v = -INFINITY;
        if (a.temperature != 1.0f) v = v * a.temperature + 0.0f;
        lg[i] = v;
    }
    __syncthreads();

    int token = -1;
    if (!a.do_sample) {
        // argmax, first maximal index (Tensor::argmax)
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = tid; i < V; i += 256) {
            const float v = lg[i];
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off, 64);
            const int oi = __shfl_xor(bi, off, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { redf[tid >> 6] = bv; redi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w)
                if (redf[w] > bv || (redf[w] == bv && redi[w] < bi)) { bv = redf[w]; bi = redi[w]; }
            sh_tok = bi == 0x7fffffff ? 0 : bi;
        }
        __syncthreads();
        token = sh_tok;
    } else {
        // ---- top-k threshold by 4-pass radix select on order-preserving keys
        if (a.top_k > 0 && a.top_k < V) {
            unsigned prefix = 0, mask = 0;
            if (tid == 0) sh_k = a.top_k;
            for (int pass = 3; pass >= 0; --pass) {
                const int shift = pass * 8;
                hist[tid] = 0;
                __syncthreads();
                for (int i = tid; i < V; i += 256) {
                    const unsigned k = okey(lg[i]);
                    if ((k & mask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1);
                }
                __syncthreads();
                if (tid < 64) {
                    // wave 0: lane l owns bins 255-4l .. 252-4l (descending); suffix counts via an inclusive wave scan
                    const int b0 = 255 - 4 * tid;
                    const int h0 = hist[b0], h1 = hist[b0 - 1], h2 = hist[b0 - 2], h3 = hist[b0 - 3];
                    const int mine = h0 + h1 + h2 + h3;
                    int incl = mine;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off, 64);
                        if (tid >= off) incl += o;
                    }
                    const int before = incl - mine;  // candidates in strictly higher bins than this lane's
                    const int kk = sh_k;
                    if (before < kk && incl >= kk) {  // exactly one lane holds the k-th largest
                        int cum = before, sel = b0;
                        if (cum + h0 >= kk) sel = b0;
                        else if ((cum += h0) + h1 >= kk) sel = b0 - 1;
                        else if ((cum += h1) + h2 >= kk) sel = b0 - 2;
                        else { cum += h2; sel = b0 - 3; }
                        sh_sel = sel;
                        sh_k = kk - cum;
                    }
                }
                __syncthreads();
                prefix |= ((unsigned)sh_sel) << shift;
                mask |= 255u << shift;
                __syncthreads();
            }
            const float thr = okey_inv(prefix);
            for (int i = tid; i < V; i += 256)
                if (lg[i] < thr) lg[i] = -INFINITY;
            __syncthreads();
        }
        // ---- compact the finite candidates in index order
        const int per = (V + 255) / 256;
        const int i0 = tid * per, i1 = min(V, i0 + per);
        int c = 0;
        for (int i = i0; i < i1; ++i) c += lg[i] > -INFINITY ? 1 : 0;
        {
            int incl = c;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off, 64);
                if ((tid & 63) >= off) incl += o;
            }
            if ((tid & 63) == 63) redi[tid >> 6] = incl;
            __syncthreads();
            int base = 0;
            for (int w = 0; w < (tid >> 6); ++w) base += redi[w];
            cnt[tid] = base + incl - c;
            if (tid == 255) sh_n = base + incl;
        }
        __syncthreads();
        {
            int w = cnt[tid];
            for (int i = i0; i < i1; ++i)
                if (lg[i] > -INFINITY) { sv[w] = lg[i]; si[w] = (unsigned short)i; ++w; }
        }
        __syncthreads();
        const int n = sh_n;
        if (n == 0) {
            token = 0;  // every logit is -inf: softmax is NaN in the reference; fall back to index 0
        } else {
            // ---- top-p
            if (a.top_p < 1.0f) {
                if (tid == 0) {
                    float mx = -INFINITY;
                    for (int i = 0; i < n; ++i) mx = fmaxf(mx, sv[i]);
                    sh_f = mx;
                }
                __syncthreads();
                const float mx = sh_f;
                __syncthreads();
                if (tid == 0) {
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += expf(sv[i] - mx);
                    sh_f = sum;
                }
                __syncthreads();
                const float sum = sh_f;
                // probabilities of the survivors (lg is free after compaction), then a stable descending rank
                for (int i = tid; i < n; i += 256) lg[i] = expf(sv[i] - mx) / sum;
                __syncthreads();
                for (int i = tid; i < n; i += 256) {
                    const float pi = lg[i];
                    int rank = 0;
                    for (int j = 0; j < n; ++j) {
                        const float pj = lg[j];
                        rank += (pj > pi || (pj == pi && j < i)) ? 1 : 0;
                    }
                    order[rank] = (unsigned short)i;
                }
                __syncthreads();
                if (tid == 0) {
                    float cum = 0.f;
                    int cutoff = n;
                    for (int r = 0; r < n; ++r) {
                        cum += lg[order[r]];
                        if (cum > a.top_p) { cutoff = r + 1; break; }
                    }
                    sh_k = cutoff;
                }
                __syncthreads();
                const int cutoff = sh_k;
                for (int r = cutoff + tid; r < n; r += 256) sv[order[r]] = -INFINITY;
                __syncthreads();
            }
            // ---- softmax over the kept candidates, then the renormalised inverse-CDF draw
            if (n <= 64) {
                // Common case (top-k <= 64): one candidate per lane of wave 0.  The reference's sequential f32 sums are walked
                // in index order with lane reads instead of LDS round trips; the arithmetic is the same, value for value.
                if (tid < 64) {
                    const bool in = tid < n;
                    const float x = in ? sv[tid] : -INFINITY;
                    const float mx = wave_max(x);
                    const float ex = in ? expf(x - mx) : 0.f;
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(ex), i));
                    const float pr = in ? ex / sum : 0.f;
                    float total = 0.f;
                    for (int i = 0; i < n; ++i) total += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr), i));
                    // Iterator::max_by keeps the LAST of equal maxima (fallbacks at sampling.rs:370-375, 391-396)
                    const float pmax = wave_max(in ? pr : -INFINITY);
                    const unsigned long long eq = __ballot(in && pr == pmax);
                    const int best = eq ? 63 - __builtin_clzll(eq) : 0;
                    int pick = -1;
                    if (total > 0.0f && isfinite(total)) {
                        const float u = a.uniforms[((long)b * (a.row_stride / a.G) + step) * a.G + a.g];
                        const float inv = 1.0f / total;
                        float cum = 0.f;
                        for (int i = 0; i < n; ++i) {
                            cum += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pr), i)) * inv;
                            if (u < cum) { pick = i; break; }
                        }
                    }
                    if (pick < 0) pick = best;
                    if (tid == 0) sh_tok = si[pick];
                }
                __syncthreads();
                token = sh_tok;
            } else {
            {
                float mloc = -INFINITY;
                for (int i = tid; i < n; i += 256) mloc = fmaxf(mloc, sv[i]);
                mloc = wave_max(mloc);
                if ((tid & 63) == 0) redf[tid >> 6] = mloc;
                __syncthreads();
                const float mx = fmaxf(fmaxf(redf[0], redf[1]), fmaxf(redf[2], redf[3]));
                for (int i = tid; i < n; i += 256) lg[i] = expf(sv[i] - mx);   // lg is free after compaction
                __syncthreads();
                if (tid == 0) {
                    float sum = 0.f;
                    for (int i = 0; i < n; ++i) sum += lg[i];
                    sh_f = sum;
                }
                __syncthreads();
                const float sum = sh_f;
                for (int i = tid; i < n; i += 256) sv[i] = lg[i] / sum;
                __syncthreads();
            }
            if (tid == 0) {
                float total = 0.f;
                for (int i = 0; i < n; ++i) total += sv[i];
                int tok = -1;
                // Iterator::max_by keeps the LAST of equal maxima (fallbacks at sampling.rs:370-375, 391-396)
                int best = 0;
                for (int i = 1; i < n; ++i)
                    if (sv[i] >= sv[best]) best = i;
                if (!(total > 0.0f) || !isfinite(total)) {
                    tok = si[best];
                } else {
                    const float u = a.uniforms[((long)b * (a.row_stride / a.G) + step) * a.G + a.g];
                    const float inv = 1.0f / total;
                    float cum = 0.f;
                    for (int i = 0; i < n; ++i) {
                        cum += sv[i] * inv;
                        if (u < cum) { tok = si[i]; break; }
                    }
                    if (tok < 0) tok = si[best];
                }
                sh_tok = tok;
            }
            __syncthreads();
            token = sh_tok;
            }
        }
    }
    // the sampled code's embedding row is the next code-predictor pass's input: copy it here instead of in a kernel of its own
    if (a.next_table) {
        const int H = a.next_H;
        if (a.next_table_bf16) {
            const bf16_t* src = (const bf16_t*)a.next_table + (long)token * H;
            if (a.next_dst_bf16) {
                bf16_t* dst = (bf16_t*)a.next_dst + (long)b * H;
                for (int i = tid * 8; i < H; i += 256 * 8) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
            } else {
                float* dst = (float*)a.next_dst + (long)b * H;
                for (int i = tid * 8; i < H; i += 256 * 8) {
                    float v[8];
                    load8(src + i, v);
                    store8(dst + i, v);
                }
            }
        } else {
            const float* src = (const float*)a.next_table + (long)token * H;
            float* dst = (float*)a.next_dst + (long)b * H;
            for (int i = tid * 4; i < H; i += 256 * 4) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(src + i);
        }
    }
    if (tid == 0 && live) {
        a.codes_base[(long)b * a.row_stride + (long)step * a.G + a.g] = (unsigned)token;
        if (a.g == 0) {
            if (a.hist_count) a.hist_count[(long)b * V + token] += 1;
            if (a.eos_step && a.eos_id >= 0 && token == a.eos_id && a.eos_step[b] < 0) a.eos_step[b] = step;
        }
    }
}

void launch_sample(hipStream_t st, const SampleArgs& a) {
    QTTS_CHECK(a.V >= 1 && a.V <= SAMPLE_MAX_V, ERR_SHAPE, "sample: vocabulary larger than the LDS staging buffer");
    QTTS_CHECK(a.B >= 1 && a.G >= 1 && a.g >= 0 && a.g < a.G && a.row_stride % a.G == 0 && a.max_steps >= 1 &&
                   (long)a.max_steps * a.G <= a.row_stride, ERR_SHAPE, "sample: bad shape");
    QTTS_CHECK(!a.do_sample || a.uniforms != nullptr, ERR_BAD_ARG, "sample: uniforms required when sampling");
    QTTS_CHECK(a.next_table == nullptr || (a.next_dst != nullptr && a.next_H % 8 == 0 && (a.next_table_bf16 || !a.next_dst_bf16)), ERR_BAD_ARG,
               "sample: bad next-row description");
    hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace qtts
