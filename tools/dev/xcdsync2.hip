// Probe 2: hand-over between the workgroups of one XCD through per-workgroup flags (no atomic RMW).
//   flag store / flag poll / payload read flavours are picked by template parameters:
//   FS: 0 plain store, 1 device-scope store (sc1)
//   FP: 0 poll with `global_load sc0`, 1 poll with `global_load sc1`, 2 poll with `global_load sc0 sc1`, 3 plain load after buffer_inv sc0, 4 plain load after buffer_inv sc1
//   INV: what is issued after the wait and before the payload reads: 0 nothing, 1 buffer_inv sc0, 2 buffer_inv sc1
//   payload buffers alternate between two sets, so a stale L1 line shows as a wrong value.
// Every spin is bounded.
// build: hipcc -O3 --offload-arch=gfx950 tools/dev/xcdsync2.hip -o tools/dev/xcdsync2
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

constexpr int PAY = 256;
constexpr unsigned SPIN_MAX = 1u << 18;

template <int FP>
__device__ __forceinline__ unsigned poll_load(const unsigned* p) {
    unsigned v;
    if (FP == 0) asm volatile("global_load_dword %0, %1, off sc0\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (FP == 1) asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (FP == 2) asm volatile("global_load_dword %0, %1, off sc0 sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (FP == 3) asm volatile("buffer_inv sc0\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("buffer_inv sc1\n global_load_dword %0, %1, off\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// state[0]: fail flag; flags[b * 16]: round counter of workgroup b
template <int FS, int FP, int INV, int GRID = 0, int PS = 0, int PL = 0>
__global__ __launch_bounds__(256) void rounds_kernel(unsigned* state, unsigned* flags, unsigned* pay, int rounds, unsigned* bad, unsigned long long* ticks,
                                                     int group) {
    const unsigned b = blockIdx.x, nb = gridDim.x;
    const unsigned x = b & 7u, slot = b >> 3, sib = nb / 8;
    unsigned wrong = 0;
    __shared__ unsigned dead;
    if (threadIdx.x == 0) dead = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (int r = 0; r < rounds; ++r) {
        unsigned* mine = pay + ((size_t)(r & 1) * nb + b) * PAY;
        if (PS == 0) mine[threadIdx.x] = (unsigned)r * 1000003u + b * 7u + threadIdx.x;
        else asm volatile("global_store_dword %0, %1, off sc1" ::"v"(mine + threadIdx.x), "v"((unsigned)r * 1000003u + b * 7u + threadIdx.x) : "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
        __syncthreads();
        if (threadIdx.x == 0) {
            if (FS == 0) asm volatile("global_store_dword %0, %1, off\n s_waitcnt vmcnt(0)" ::"v"(flags + b * 16), "v"((unsigned)(r + 1)) : "memory");
            else asm volatile("global_store_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" ::"v"(flags + b * 16), "v"((unsigned)(r + 1)) : "memory");
        }
        if (threadIdx.x < (GRID ? nb : sib)) {  // lane s polls sibling s of this XCD (or workgroup s of the grid)
            const unsigned* f = flags + (GRID ? threadIdx.x : threadIdx.x * 8u + x) * 16;
            unsigned spins = 0;
            while (poll_load<FP>(f) < (unsigned)(r + 1)) {
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
        if (INV == 1) asm volatile("buffer_inv sc0" ::: "memory");
        if (INV == 2) asm volatile("buffer_inv sc1" ::: "memory");
        const unsigned ng = group > 0 ? (unsigned)group : sib;
        for (unsigned s = 0; s < ng; ++s) {
            const unsigned ob = GRID ? (b + 1 + s * 3) % nb : ((slot + s) % sib) * 8u + x;
            const unsigned* theirs = pay + ((size_t)(r & 1) * nb + ob) * PAY;
            unsigned got;
            if (PL == 0) got = theirs[threadIdx.x];
            else asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(got) : "v"(theirs + threadIdx.x) : "memory");
            wrong += got != (unsigned)r * 1000003u + ob * 7u + threadIdx.x;
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (wrong) atomicAdd(bad, wrong);
    if (threadIdx.x == 0) ticks[b] = t1 - t0;
}

template <int FS, int FP, int INV, int GRID = 0, int PS = 0, int PL = 0>
static void run(const char* name, int nb, int rounds, int group) {
    unsigned *state, *flags, *pay, *bad;
    unsigned long long* ticks;
    CK(hipMalloc(&state, 4096));
    CK(hipMalloc(&flags, (size_t)nb * 64));
    CK(hipMalloc(&pay, (size_t)2 * nb * PAY * 4));
    CK(hipMalloc(&bad, 4));
    CK(hipMalloc(&ticks, (size_t)nb * 8));
    for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemset(state, 0, 4096));
        CK(hipMemset(flags, 0, (size_t)nb * 64));
        CK(hipMemset(bad, 0, 4));
        CK(hipMemset(pay, 0xff, (size_t)2 * nb * PAY * 4));
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((rounds_kernel<FS, FP, INV, GRID, PS, PL>), dim3(nb), dim3(256), 0, 0, state, flags, pay, rounds, bad, ticks, group);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        unsigned hbad = 0, hstate[1];
        CK(hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(hstate, state, 4, hipMemcpyDeviceToHost));
        printf("%-44s blocks %4d group %3d rounds %5d: %8.3f us/round, wrong %u, timed_out %u\n", name, nb, group, rounds, ms * 1e3 / rounds, hbad,
               hstate[0]);
        fflush(stdout);
        if (hstate[0]) break;
    }
    CK(hipFree(state));
    CK(hipFree(flags));
    CK(hipFree(pay));
    CK(hipFree(bad));
    CK(hipFree(ticks));
}

int main() {
    run<0, 1, 0, 0, 0, 1>("XCD: plain stores, sc1 poll, sc1 payload loads", 256, 2000, 4);
    run<0, 1, 0, 0, 0, 1>("XCD: plain stores, sc1 poll, sc1 payload loads", 256, 2000, 0);
    run<0, 1, 0, 0, 0, 1>("XCD: plain stores, sc1 poll, sc1 payload loads", 512, 2000, 4);
    run<0, 1, 0, 0, 0, 1>("XCD: plain stores, sc1 poll, sc1 payload loads", 768, 2000, 4);
    run<1, 1, 0, 1, 1, 1>("grid: sc1 stores, sc1 poll, sc1 payload loads", 256, 2000, 4);
    run<1, 1, 0, 1, 1, 1>("grid: sc1 stores, sc1 poll, sc1 payload loads", 256, 2000, 16);
    run<0, 1, 0, 1, 0, 1>("grid: plain stores, sc1 poll, sc1 payload loads", 256, 2000, 4);
    return 0;
}
