// Probe: can the workgroups that share one XCD's L2 hand data to each other inside a kernel cheaply?
//   1. which XCD does workgroup b of a 1-D grid run on (hardware register XCC_ID against b % 8)
//   2. rounds of [store a payload -> arrive on a counter -> wait for the 32 siblings -> read their payloads]
//      mode 0: siblings = the workgroups of the same XCD; plain stores (they land in that XCD's L2), counter bumped by an
//              L2 atomic, polled by a returning L2 atomic, L1 invalidated before the payload reads
//      mode 1: all workgroups of the grid; device-scope stores / atomics / loads (no cache write-back or invalidate)
//      mode 2: no hand-over at all (loop overhead)
// Every spin is bounded; a timed-out wait raises a flag that ends every loop.
// build: hipcc -O3 --offload-arch=gfx950 tools/dev/xcdsync.hip -o tools/dev/xcdsync
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u; }

__global__ void where_kernel(unsigned* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = xcc_id();
}

constexpr int PAY = 256;  // dwords of payload per workgroup and round
constexpr unsigned SPIN_MAX = 1u << 22;

// state: [0] fail flag, [16 + 16 x] counter of XCD x (mode 0) / [16] the grid's counter (mode 1)
template <int MODE>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* state, unsigned* pay, int rounds, unsigned* bad, unsigned long long* ticks, int group) {
    const unsigned b = blockIdx.x, nb = gridDim.x;
    const unsigned x = b & 7u, slot = b >> 3;           // XCD and index inside it
    unsigned* cnt = MODE == 0 ? state + 16 + 16 * x : state + 16;
    const unsigned sib = MODE == 0 ? nb / 8 : nb;       // workgroups that meet
    unsigned wrong = 0;
    __shared__ unsigned dead;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        // payload of this round: two buffers in turn, so that a fast sibling's next round does not overwrite what a slow one reads
        unsigned* mine = pay + ((size_t)(r & 1) * nb + b) * PAY;
        const unsigned v = (unsigned)r * 1000003u + b * 7u + threadIdx.x;
        if (MODE == 1) __hip_atomic_store(mine + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else mine[threadIdx.x] = v;
        if (MODE != 2) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the stores are acknowledged
            __syncthreads();
            if (threadIdx.x == 0) {
                if (MODE == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned want = sib * (unsigned)(r + 1);
                unsigned spins = 0;
                for (;;) {
                    unsigned c;
                    if (MODE == 0) c = __hip_atomic_fetch_or(cnt, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    else c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (c >= want) break;
                    if (++spins > SPIN_MAX || __hip_atomic_load(state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        __hip_atomic_store(state, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (dead) break;
            if (MODE == 0) asm volatile("buffer_inv sc0" ::: "memory");
            // read `group` siblings' payloads (all of them when group == 0)
            const unsigned ng = group > 0 ? (unsigned)group : sib;
            for (unsigned s = 0; s < ng; ++s) {
                const unsigned ob = MODE == 0 ? ((slot + s) % sib) * 8u + x : (b + s) % nb;
                const unsigned* theirs = pay + ((size_t)(r & 1) * nb + ob) * PAY;
                unsigned got;
                if (MODE == 1) got = __hip_atomic_load(theirs + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else got = theirs[threadIdx.x];
                wrong += got != (unsigned)r * 1000003u + ob * 7u + threadIdx.x;
            }
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) ticks[b] = t1 - t0;
}

template <int MODE>
static void run(const char* name, int nb, int rounds, int group) {
    unsigned *state, *pay, *bad;
    unsigned long long* ticks;
    CK(hipMalloc(&state, 4096));
    CK(hipMalloc(&pay, (size_t)2 * nb * PAY * 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&ticks, (size_t)nb * 8));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemset(state, 0, 4096));
        CK(hipMemset(bad, 0, 4));
        CK(hipMemset(pay, 0xff, (size_t)2 * nb * PAY * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((rounds_kernel<MODE>), dim3(nb), dim3(256), 0, 0, state, pay, rounds, bad, ticks, group);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hbad = 0, hstate[1];
        std::vector<unsigned long long> ht((size_t)nb);
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hstate, state, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ht.data(), ticks, (size_t)nb * 8, hipMemcpyDeviceToHost));
        unsigned long long tmax = 0;
        for (auto t : ht) tmax = t > tmax ? t : tmax;
        printf("%-28s blocks %4d group %3d rounds %5d: %8.3f us/round (event), %8.3f us/round (clock, 100 MHz), wrong %u, timed_out %u\n", name, nb, group,
               rounds, ms * 1e3 / rounds, (double)tmax * 0.01 / rounds, hbad, hstate[0]);
        fflush(stdout);
        if (hstate[0]) break;
    }
    CK(hipFree(state));
    CK(hipFree(pay));
    CK(hipFree(bad));
    CK(hipFree(ticks));
}

int main() {
    for (int nb : {256, 512, 1024}) {
        unsigned* out;
        CK(hipMalloc(&out, (size_t)nb * 4));
        hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(64), 0, 0, out);
        CK(hipDeviceSynchronize());
        std::vector<unsigned> h((size_t)nb);
        CK(hipMemcpy(h.data(), out, (size_t)nb * 4, hipMemcpyDeviceToHost));
        int mism = 0;
        for (int b = 0; b < nb; ++b) mism += h[(size_t)b] != (unsigned)(b % 8);
        printf("grid %4d: workgroups whose XCC_ID differs from id %% 8: %d  (first 16:", nb, mism);
        for (int b = 0; b < 16; ++b) printf(" %u", h[(size_t)b]);
        printf(")\n");
        CK(hipFree(out));
    }
    run<2>("no hand-over", 256, 2000, 0);
    run<0>("same-XCD hand-over", 256, 2000, 0);
    run<0>("same-XCD hand-over", 256, 2000, 4);
    run<0>("same-XCD hand-over", 512, 2000, 4);
    run<1>("device-scope hand-over", 256, 2000, 4);
    run<1>("device-scope hand-over", 256, 2000, 0);
    return 0;
}
