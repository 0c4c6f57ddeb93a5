// On-device token sampling: the reference's logit processors in their HF order and its inverse-CDF
// draw (qwen_tts/src/nn/sampling.rs:310-397), one 256-thread block per batch row, logits staged in LDS.
//   repetition penalty (applied once per earlier occurrence, :184-193) -> suppress range / EOS while
//   history < min_new_tokens (:319-339) -> * 1/T (:199-204) -> top-k: keep >= k-th largest (:209-229)
//   -> top-p: stable sort by prob, shortest prefix with cumsum > p (:234-274) -> argmax, or softmax +
//   renormalised running sum against one 24-bit uniform (:356-397).
// The sequential f32 sums of the reference are reproduced exactly: with at most 64 surviving candidates (the usual top-k) a
// wave holds one per lane in index order and a chain of shift-by-one-lane adds builds the left-to-right running sums; larger
// sets are walked by a single lane.  Adding the exact zeros of removed candidates would not change them.
// top-k (k <= 64) takes two steps instead of a radix select over the whole vocabulary: the k-th largest of 64 group maxima
// is a lower bound of the k-th largest logit that only a few dozen candidates pass, and the exact threshold, the kept set
// and its index order come from rank counting among those.  Any other case (or more than 256 candidates above the bound)
// goes through the radix select.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

constexpr int SAMPLE_MAX_V = 4096;
constexpr int SAMPLE_CAND = 256;   // most candidates the two-step top-k handles

__device__ __forceinline__ unsigned okey(float f) {  // order-preserving map float -> uint
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float okey_inv(unsigned k) {
    const unsigned u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// Left-to-right running sum over the wave's lanes with the association of a sequential loop, ((x0 + x1) + x2) + ...: lane i
// ends with the sum of lanes 0..i.  Each step adds the left neighbour's value (a shift by one lane, zero into lane 0); after
// step t the lanes 0..t are final, and recomputing a final lane from its final neighbour changes nothing.
__device__ __forceinline__ float lane_running_sum(float x) {
    float c = x;
#pragma unroll
    for (int t = 1; t < 64; ++t)
        c = x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, c), 0x138 /* wave_shr:1 */, 0xf, 0xf, true));
    return c;
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
    __shared__ unsigned gkey[64];        // group maxima as order-preserving keys, low 6 bits = group (unique)
    __shared__ float cv[SAMPLE_CAND];    // candidates above the lower bound: value, vocabulary index (unordered)
    __shared__ int ci[SAMPLE_CAND];
    __shared__ int kidx[64 + 4];         // vocabulary indices of the kept candidates (unordered)
    __shared__ int sh_n1, sh_n2;

    QTTS_TS(0);
    const int b = blockIdx.x, tid = threadIdx.x, V = a.V;
    // a retired row (active[b] == 0) still runs - the launch covers every slot - on its last frame index, and writes nothing
    const bool live = a.active == nullptr || a.active[b] != 0;
    const int step = min(a.step[b], a.max_steps - 1);
    const float* src = a.logits + (long)b * V;
    const int* hc = a.hist_count ? a.hist_count + (long)b * V : nullptr;
    const bool eos_blocked = a.eos_id >= 0 && step < a.min_new_tokens;
    // the draw's uniform number is known from the start: requested now, used at the very end
    const float u_draw = a.do_sample ? a.uniforms[((long)b * (a.row_stride / a.G) + step) * a.G + a.g] : 0.f;
    const bool two_step = a.do_sample && a.top_k > 0 && a.top_k < V && a.top_k <= 64;
    float tmax = -INFINITY;
    cv[tid] = -INFINITY;
    if (tid < 64 + 4) kidx[tid] = 0x7fffffff;
    if (tid == 0) { sh_n1 = 0; sh_n2 = 0; }

    // every thread keeps the logits it stages (element tid + 256 j) in registers: the loads travel together, and the
    // candidate search below needs no second look at LDS
    constexpr int PER = SAMPLE_MAX_V / 256;
    float vals[PER];
    const bool penal = hc && a.rep_penalty != 1.0f;
    int occ[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {  // branch-free requests (an index past the end repeats the last element): all in flight together
        vals[j] = 0.f;
        occ[j] = 0;
        if (256 * j < V) {  // (the same for every thread)
            const int i = min(tid + 256 * j, V - 1);
            vals[j] = src[i];
            if (penal) occ[j] = hc[i];
        }
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int i = tid + 256 * j;
        if (i < V) {
            float v = vals[j];
            for (int r = 0; r < occ[j]; ++r) v = v > 0.0f ? v / a.rep_penalty : v * a.rep_penalty;
            if ((i >= a.suppress_lo && i < a.suppress_hi && i != a.eos_id) || (eos_blocked && i == a.eos_id)) v = -INFINITY;
            if (a.temperature != 1.0f) v = v * a.temperature + 0.0f;
            lg[i] = v;
            vals[j] = v;
            tmax = fmaxf(tmax, v);
        } else {
            vals[j] = -INFINITY;
        }
    }
    if (two_step) {
        // maximum of each group of four threads (quad permutes), one key per group
        tmax = fmaxf(tmax, dpp_move<0xB1>(tmax));  // quad_perm [1,0,3,2]
        tmax = fmaxf(tmax, dpp_move<0x4E>(tmax));  // quad_perm [2,3,0,1]
        if ((tid & 3) == 0) gkey[tid >> 2] = (okey(tmax) & ~63u) | (unsigned)(tid >> 2);
    }
    __syncthreads();
    QTTS_TS(1);

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
        bool compacted = false;
        if (two_step) {
            // ---- lower bound: the k-th largest group maximum (every wave ranks the 64 unique keys for itself)
            const int lane = tid & 63;
            const unsigned mykey = gkey[lane];
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 64; j += 4) {
                const uint4 q = *reinterpret_cast<const uint4*>(&gkey[j]);
                rank += (q.x > mykey ? 1 : 0) + (q.y > mykey ? 1 : 0) + (q.z > mykey ? 1 : 0) + (q.w > mykey ? 1 : 0);
            }
            const unsigned long long hit = __ballot(rank == a.top_k - 1);
            const unsigned bound = (unsigned)__builtin_amdgcn_readlane((int)mykey, (int)__builtin_ctzll(hit)) & ~63u;
            QTTS_TS(7);
            // ---- candidates: every thread looks through the logits it staged itself and reserves room for its own with one add
            int mine_n = 0;
#pragma unroll
            for (int j = 0; j < PER; ++j) mine_n += (tid + 256 * j < V && okey(vals[j]) >= bound) ? 1 : 0;
            if (mine_n) {
                int p = atomicAdd(&sh_n1, mine_n);
#pragma unroll
                for (int j = 0; j < PER; ++j) {
                    if (tid + 256 * j < V && okey(vals[j]) >= bound) {
                        if (p < SAMPLE_CAND) { cv[p] = vals[j]; ci[p] = tid + 256 * j; }
                        ++p;
                    }
                }
            }
            __syncthreads();
            QTTS_TS(8);
            const int n1 = sh_n1;
            if (n1 <= SAMPLE_CAND) {
                // ---- kept = fewer than k candidates are strictly larger (exactly the candidates >= the k-th largest logit)
                const float mine = cv[tid];
                const int my_i = ci[tid];
                bool kept = false;
                if (tid < n1) {
                    int cg = 0;
                    for (int j0 = 0; j0 < n1; j0 += 32) {   // entries past n1 are -inf; 32 at a time keeps eight reads in flight
#pragma unroll
                        for (int j = 0; j < 32; j += 4) {
                            const float4 q = *reinterpret_cast<const float4*>(&cv[j0 + j]);
                            cg += (q.x > mine ? 1 : 0) + (q.y > mine ? 1 : 0) + (q.z > mine ? 1 : 0) + (q.w > mine ? 1 : 0);
                        }
                    }
                    kept = cg < a.top_k && mine > -INFINITY;
                    if (kept) {
                        const int q = atomicAdd(&sh_n2, 1);
                        if (q < 64) kidx[q] = my_i;
                    }
                }
                __syncthreads();
                QTTS_TS(9);
                const int n2 = sh_n2;
                if (n2 <= 64) {  // (ties at the threshold can keep more than k: then the radix path sorts it out)
                    if (kept) {
                        int pos = 0;
#pragma unroll
                        for (int j = 0; j < 64; j += 4) {   // entries past n2 are INT_MAX
                            const int4 q = *reinterpret_cast<const int4*>(&kidx[j]);
                            pos += (q.x < my_i ? 1 : 0) + (q.y < my_i ? 1 : 0) + (q.z < my_i ? 1 : 0) + (q.w < my_i ? 1 : 0);
                        }
                        sv[pos] = mine;
                        si[pos] = (unsigned short)my_i;
                    }
                    if (tid == 0) sh_n = n2;
                    compacted = true;
                    __syncthreads();
                }
            }
        }
        if (!compacted) {
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
            QTTS_TS(2);
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
        }
        const int n = sh_n;
        QTTS_TS(3);
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
            QTTS_TS(4);
            // ---- softmax over the kept candidates, then the renormalised inverse-CDF draw
            if (n <= 64) {
                // Common case (top-k <= 64): one candidate per lane of wave 0.  The reference's sequential f32 sums are walked
                // in index order with lane reads instead of LDS round trips; the arithmetic is the same, value for value.
                if (tid < 64) {
                    const bool in = tid < n;
                    const float x = in ? sv[tid] : -INFINITY;
                    const float mx = wave_max(x);
                    const float ex = in ? expf(x - mx) : 0.f;
                    const float sum = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_running_sum(ex)), 63));
                    const float pr = in ? ex / sum : 0.f;
                    const float total = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lane_running_sum(pr)), 63));
                    // Iterator::max_by keeps the LAST of equal maxima (fallbacks at sampling.rs:370-375, 391-396)
                    const float pmax = wave_max(in ? pr : -INFINITY);
                    const unsigned long long eq = __ballot(in && pr == pmax);
                    const int best = eq ? 63 - __builtin_clzll(eq) : 0;
                    int pick = -1;
                    if (total > 0.0f && isfinite(total)) {
                        const float cum = lane_running_sum(pr * (1.0f / total));
                        const unsigned long long below = __ballot(in && u_draw < cum);
                        if (below) pick = (int)__builtin_ctzll(below);
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
                    const float u = u_draw;
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
    QTTS_TS(5);
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
    if (a.memo_x) {
        const float4* sx = reinterpret_cast<const float4*>(a.memo_x + (long)token * a.memo_Hx);
        float4* dx = reinterpret_cast<float4*>(a.memo_x_dst + (long)b * a.memo_Hx);
        for (int i = tid; i < a.memo_Hx / 4; i += 256) dx[i] = sx[i];
        const float4* sq = reinterpret_cast<const float4*>(a.memo_q + (long)token * a.memo_Nq);
        float4* dq = reinterpret_cast<float4*>(a.memo_q_dst + (long)b * a.memo_Nq);
        for (int i = tid; i < a.memo_Nq / 4; i += 256) dq[i] = sq[i];
    }
    if (tid == 0 && live) {
        a.codes_base[(long)b * a.row_stride + (long)step * a.G + a.g] = (unsigned)token;
        if (a.g == 0) {
            if (a.hist_count) a.hist_count[(long)b * V + token] += 1;
            if (a.eos_step && a.eos_id >= 0 && token == a.eos_id && a.eos_step[b] < 0) a.eos_step[b] = step;
        }
    }
    QTTS_TS(6);
}

void launch_sample(hipStream_t st, const SampleArgs& a) {
    QTTS_CHECK(a.V >= 1 && a.V <= SAMPLE_MAX_V, ERR_SHAPE, "sample: vocabulary larger than the LDS staging buffer");
    QTTS_CHECK(a.B >= 1 && a.G >= 1 && a.g >= 0 && a.g < a.G && a.row_stride % a.G == 0 && a.max_steps >= 1 &&
                   (long)a.max_steps * a.G <= a.row_stride, ERR_SHAPE, "sample: bad shape");
    QTTS_CHECK(!a.do_sample || a.uniforms != nullptr, ERR_BAD_ARG, "sample: uniforms required when sampling");
    QTTS_CHECK(a.memo_x == nullptr || (a.memo_x_dst && a.memo_q && a.memo_q_dst && a.memo_Hx % 4 == 0 && a.memo_Nq % 4 == 0), ERR_BAD_ARG,
               "sample: bad tabulated-row description");
    QTTS_CHECK(a.next_table == nullptr || (a.next_dst != nullptr && a.next_H % 8 == 0 && (a.next_table_bf16 || !a.next_dst_bf16)), ERR_BAD_ARG,
               "sample: bad next-row description");
    hipLaunchKernelGGL(sample_kernel, dim3(a.B), dim3(256), 0, st, a);
}

}  // namespace qtts
