// Diagnostics only: pure HBM read streams with the access shapes the GEMM kernels use, to measure what a
// kernel of a given size can reach at best (launch ramp, tail and access-pattern effects included).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// pattern 0: each wave instruction reads 1 KiB contiguous (lane * 16 B)
// pattern 1: each wave instruction reads 16 rows x 64 B (row stride `ld` bytes), two instructions per 128-B span
template <int PATTERN, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, unsigned* sink, long n_pieces, int ld16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (PATTERN == 0) {
        // piece = UNROLL KiB contiguous; pieces are dealt round-robin to waves (every index < n_pieces * 64 * UNROLL)
        for (long p = wid; p < n_pieces; p += nw) {
            const uint4* base = src + p * (64 * UNROLL);
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = base[u * 64 + lane];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    } else {
        // piece = 16 rows x (UNROLL * 128) bytes: one unrolled burst of 16-row x 64-byte wave loads
        const int r = lane & 15, g = lane >> 4;
        const int segs = ld16 / (8 * UNROLL);
        for (long p = wid; p < n_pieces; p += nw) {
            const long rg = p / segs;
            const int c = (int)(p % segs) * 8 * UNROLL;
            const uint4* rowp = src + (rg * 16 + r) * ld16 + c;
            uint4 v[2 * UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                v[2 * u] = rowp[u * 8 + g * 2];
                v[2 * u + 1] = rowp[u * 8 + g * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < 2 * UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

void launch_stream(hipStream_t st, const void* src, unsigned* sink, long bytes, int blocks, int pattern, int unroll, int ld_bytes) {
    const int U = unroll >= 8 ? 8 : 4;
    QTTS_CHECK(blocks >= 1 && blocks <= 65535, ERR_SHAPE, "stream: bad block count");
    long pieces;
    if (pattern == 0) {
        QTTS_CHECK(bytes % (1024L * U) == 0, ERR_SHAPE, "stream: bytes must be a multiple of the unrolled KiB piece");
        pieces = bytes / (1024L * U);
    } else {
        QTTS_CHECK(ld_bytes % (128 * U) == 0 && bytes % ((long)ld_bytes * 16) == 0, ERR_SHAPE,
                   "stream: row length must be a multiple of the unrolled 128-byte step and rows a multiple of 16");
        pieces = bytes / ((long)ld_bytes * 16) * (ld_bytes / (128 * U));
    }
    const int ld16 = ld_bytes / 16;
    if (pattern == 0) {
        if (U == 8) hipLaunchKernelGGL((stream_kernel<0, 8>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
        else hipLaunchKernelGGL((stream_kernel<0, 4>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
    } else {
        if (U == 8) hipLaunchKernelGGL((stream_kernel<1, 8>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
        else hipLaunchKernelGGL((stream_kernel<1, 4>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
    }
}


// ------------------------------------------------------------------------------------------------
// Grid-wide barrier cost: `iters` barriers inside one resident grid.  Every block publishes a value before each
// barrier and checks its neighbour's afterwards (errors are counted, not trapped).  The spin is bounded and an abort
// flag is raised when a wait runs out, so that every wave reaches the end of the kernel whatever happens.
__device__ __forceinline__ bool bench_barrier(unsigned* counter, unsigned* abort_flag, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 4000000L || __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    return __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u;
}

__global__ void barrier_bench_kernel(unsigned* counter, unsigned* abort_flag, unsigned* slots, unsigned* errors, int iters) {
    const unsigned nb = gridDim.x;
    for (int it = 0; it < iters; ++it) {
        if (threadIdx.x == 0) slots[blockIdx.x] = (unsigned)it * 7919u + blockIdx.x;
        if (!bench_barrier(counter, abort_flag, (unsigned)(it + 1) * nb)) return;
        if (threadIdx.x == 0) {
            const unsigned nbr = (blockIdx.x + 1) % nb;
            // plain load: the barrier's acquire must make the neighbour's plain store visible
            if (slots[nbr] != (unsigned)it * 7919u + nbr) atomicAdd(errors, 1u);
        }
        // second barrier so that the next iteration's store cannot overtake this iteration's check
        if (!bench_barrier(counter + 32, abort_flag, (unsigned)(it + 1) * nb)) return;
    }
}

__global__ void null_kernel(unsigned* p) {
    if (p == nullptr) return;
}

// Chain of dependent kernels: out[i] = in[i] + 1 over `n` floats.  mode 0: plain stores; 1: 16-byte write-through stores
// (sc0 sc1); 2: 8-byte write-through stores.  Measures what a kernel boundary costs when the producer leaves dirty lines in
// its L2 and when it does not.
__global__ __launch_bounds__(256) void chain_kernel(const float* __restrict__ in, float* out, int n4, int mode) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        float* p = out + 4L * i;
        if (mode == 0) {
            *reinterpret_cast<float4*>(p) = v;
        } else if (mode == 1) {
            f32x4_t w = {v.x, v.y, v.z, v.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(w) : "memory");
        } else {
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(a) : "memory");
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p + 2), "v"(b) : "memory");
        }
    }
}
// a second, distinct kernel with the same body plus a large amount of (never executed) code: alternating the two in a chain
// shows what switching between different, bigger kernels costs at a boundary
__global__ __launch_bounds__(256) void chain_kernel_b(const float* __restrict__ in, float* out, int n4, int mode) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(in)[i];
        if (mode > 100) {  // dead at run time, but the compiler has to keep it: ~ tens of KB of code
#pragma unroll
            for (int r = 0; r < 512; ++r) {
                v.x = v.x * 1.0001f + __sinf(v.y + (float)r);
                v.y = v.y * 0.9999f + __cosf(v.z + (float)r);
                v.z = v.z * 1.0002f + __expf(v.w * 1e-3f);
                v.w = v.w * 0.9998f + v.x;
            }
        }
        v.x += 1.f; v.y += 1.f; v.z += 1.f; v.w += 1.f;
        *reinterpret_cast<float4*>(out + 4L * i) = v;
    }
}
void launch_chain_b(hipStream_t st, const float* in, float* out, int n, int blocks) {
    hipLaunchKernelGGL(chain_kernel_b, dim3(blocks), dim3(256), 0, st, in, out, n / 4, 0);
}
void launch_chain(hipStream_t st, const float* in, float* out, int n, int blocks, int mode) {
    QTTS_CHECK(n % 4 == 0 && n >= 4 && blocks >= 1 && blocks <= 4096 && mode >= 0 && mode <= 2, ERR_SHAPE, "chain bench: bad shape");
    hipLaunchKernelGGL(chain_kernel, dim3(blocks), dim3(256), 0, st, in, out, n / 4, mode);
}

void launch_barrier_bench(hipStream_t st, unsigned* state /* >= 64 + blocks words, zeroed */, int blocks, int threads, int iters) {
    QTTS_CHECK(blocks >= 1 && blocks <= 1024 && threads >= 64 && threads <= 1024 && iters >= 1 && iters <= 100000, ERR_SHAPE, "barrier bench: bad shape");
    hipLaunchKernelGGL(barrier_bench_kernel, dim3(blocks), dim3(threads), 0, st, state, state + 16, state + 64, state + 17, iters);
}
void launch_null(hipStream_t st) { hipLaunchKernelGGL(null_kernel, dim3(1), dim3(64), 0, st, (unsigned*)nullptr); }

}  // namespace qtts
