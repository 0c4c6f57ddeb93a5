// Dispatch-rate probe: chains of dependent launches replayed from HIP graphs, on 1..4 streams at once.
// Prints the time per launch of one stream and the launches per second of all streams together.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dev/launchbench tools/dev/launchbench.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
    do {                                                                           \
        hipError_t e_ = (x);                                                       \
        if (e_ != hipSuccess) {                                                    \
            printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
            return 1;                                                              \
        }                                                                          \
    } while (0)

__global__ void probe_kernel(float* p, int spin) {
    float v = p[threadIdx.x & 63];
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.678f) p[0] = v;  // never true; keeps the loop
}

int main(int argc, char** argv) {
    const int chain = 640, reps = 20, max_streams = argc > 1 ? atoi(argv[1]) : 4;
    float* buf;
    CK(hipMalloc(&buf, 1 << 20));
    CK(hipMemset(buf, 0, 1 << 20));
    const int grids[] = {256};
    const int spins[] = {0, 2000};
    for (int spin : spins)
        for (int grid : grids)
            for (int ns = 1; ns <= max_streams; ++ns) {
                std::vector<hipStream_t> st(ns);
                std::vector<hipGraphExec_t> ex(ns);
                for (int s = 0; s < ns; ++s) {
                    CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking));
                    hipGraph_t g;
                    CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
                    for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(probe_kernel, dim3(grid), dim3(256), 0, st[s], buf + s * 4096, spin);
                    CK(hipStreamEndCapture(st[s], &g));
                    CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
                    CK(hipGraphDestroy(g));
                }
                for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ex[s], st[s]));
                CK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < reps; ++r)
                    for (int s = 0; s < ns; ++s) CK(hipGraphLaunch(ex[s], st[s]));
                CK(hipDeviceSynchronize());
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
                printf("spin %5d grid %5d streams %d: %.2f us per launch per stream, %.2f M launches/s in all\n", spin, grid, ns,
                       us / (reps * chain), (double)reps * chain * ns / us);
                fflush(stdout);
                for (int s = 0; s < ns; ++s) {
                    CK(hipGraphExecDestroy(ex[s]));
                    CK(hipStreamDestroy(st[s]));
                }
            }
    CK(hipFree(buf));
    return 0;
}
