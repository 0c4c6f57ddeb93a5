// Cost of a kernel boundary against an in-kernel producer -> consumer hand-over through a device-scope counter (development probe).
// Chain of dependent stages; every stage reads what the previous one wrote (256 blocks, 4 KiB per block) plus a private
// "weight" block that does not depend on the producer.
//   (a) one launch per stage
//   (b) two stages per launch: blocks 0..255 produce, fence and count; blocks 256..511 request their weights, then poll the
//       counter (bounded), then consume
//   hipcc --offload-arch=gfx950 -O2 -o tools/dev/fusebench tools/dev/fusebench.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                         \
    do {                                                              \
        hipError_t e_ = (x);                                          \
        if (e_ != hipSuccess) {                                       \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                                 \
        }                                                             \
    } while (0)

constexpr int NB = 256, NT = 256;

__device__ __forceinline__ float4 stage_body(const float4* in, const float4* w, int blk) {
    const float4 a = in[(size_t)((blk * 7 + 3) % NB) * NT + threadIdx.x];  // another block's output: a real cross-block dependency
    const float4 b = w[(size_t)blk * NT + threadIdx.x];
    return make_float4(a.x + b.x, a.y + b.y, a.z * 0.5f + b.z, a.w + b.w);
}

__global__ __launch_bounds__(NT) void stage_kernel(const float4* in, float4* out, const float4* w) {
    out[(size_t)blockIdx.x * NT + threadIdx.x] = stage_body(in, w, blockIdx.x);
}

__global__ __launch_bounds__(NT) void pair_kernel(const float4* in, float4* mid, float4* out, const float4* w1, const float4* w2,
                                                  unsigned* counter, unsigned* fail) {
    if (blockIdx.x < NB) {
        mid[(size_t)blockIdx.x * NT + threadIdx.x] = stage_body(in, w1, blockIdx.x);
        __threadfence();  // the block's stores are visible device-wide before it is counted
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(counter, 1u);
    } else {
        const int blk = blockIdx.x - NB;
        const float4 b = w2[(size_t)blk * NT + threadIdx.x];  // does not depend on the producers: travels while they run
        __shared__ int ok;
        if (threadIdx.x == 0) {
            int spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NB && spins < (1 << 20)) {
                __builtin_amdgcn_s_sleep(2);
                ++spins;
            }
            ok = spins < (1 << 20);
            if (!ok) atomicAdd(fail, 1u);
        }
        __syncthreads();
        if (!ok) return;  // every wave of the block leaves
        __threadfence();
        const float4 a = mid[(size_t)((blk * 7 + 3) % NB) * NT + threadIdx.x];
        out[(size_t)blk * NT + threadIdx.x] = make_float4(a.x + b.x, a.y + b.y, a.z * 0.5f + b.z, a.w + b.w);
    }
}

int main() {
    const int stages = 200, reps = 20;
    float4 *bufA, *bufB, *bufC, *w;
    unsigned *counters, *fail;
    const size_t bytes = (size_t)NB * NT * sizeof(float4);
    CK(hipMalloc(&bufA, bytes)); CK(hipMalloc(&bufB, bytes)); CK(hipMalloc(&bufC, bytes));
    CK(hipMalloc(&w, bytes * stages));
    CK(hipMalloc(&counters, stages * sizeof(unsigned))); CK(hipMalloc(&fail, 4));
    CK(hipMemset(bufA, 0, bytes)); CK(hipMemset(w, 0, bytes * stages)); CK(hipMemset(fail, 0, 4));
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g;
        hipGraphExec_t ex;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        CK(hipMemsetAsync(counters, 0, stages * sizeof(unsigned), st));
        if (mode == 0) {
            for (int s = 0; s < stages; ++s)
                hipLaunchKernelGGL(stage_kernel, dim3(NB), dim3(NT), 0, st, (s & 1) ? bufB : bufA, (s & 1) ? bufA : bufB, w + (size_t)s * NB * NT);
        } else {
            for (int s = 0; s < stages; s += 2)
                hipLaunchKernelGGL(pair_kernel, dim3(2 * NB), dim3(NT), 0, st, ((s >> 1) & 1) ? bufC : bufA, bufB, ((s >> 1) & 1) ? bufA : bufC,
                                   w + (size_t)s * NB * NT, w + (size_t)(s + 1) * NB * NT, counters + s, fail);
        }
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        const auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ex, st));
        CK(hipStreamSynchronize(st));
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        unsigned hf = 0;
        CK(hipMemcpy(&hf, fail, 4, hipMemcpyDeviceToHost));
        printf("%s: %.2f us per stage (%u failed waits)\n", mode ? "two stages per launch" : "one launch per stage  ", us / (reps * stages), hf);
        CK(hipGraphExecDestroy(ex));
        CK(hipGraphDestroy(g));
    }
    return 0;
}
