// In-kernel timeline of the finished-output decode GEMMs (development probe): tools/gemm_harness.cpp compiled together with
// the kernel source and its wall-clock marks; prints where the first and the last block of the last launch spent their time.
//   gemm_harness <stream|full_k|k_split|slab> M N K lo(0/1) [iters] [mode: 0 plain, 1 swiglu]
// Build: hipcc -std=c++17 -O3 --offload-arch=gfx950 -I include -I qwen3-tts-rs_amd/csrc -o tools/dev/gemm_ts tools/dev/gemm_ts.hip
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define QTTS_KERNEL_TS 1
#define HARNESS_FUSED_ONLY 1
#include "../../qwen3-tts-rs_amd/csrc/k_gemm_fused.hip"

using namespace qtts;

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s <stream|full_k|k_split|slab> M N K lo [iters] [mode]\n", argv[0]);
        return 2;
    }
    const std::string kind = argv[1];
    const int M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
    const bool lo = atoi(argv[5]) != 0;
    const int iters = argc > 6 ? atoi(argv[6]) : 200;
    const int mode = argc > 7 ? atoi(argv[7]) : 0;
    try {
        const int rows = fused_rows(M);
        const size_t wbytes = (size_t)N * K * 2;
        const int copies = getenv("HARNESS_COPIES") ? atoi(getenv("HARNESS_COPIES")) : (int)std::max<size_t>(2, (size_t)600e6 / wbytes);  // default: more than the memory-side cache holds
        void *W, *X, *act, *xb;
        float *out, *x, *ss, *nw;
        QTTS_HIP(hipMalloc(&W, wbytes * copies));
        QTTS_HIP(hipMalloc(&X, (size_t)rows * K * 2 * 2));
        QTTS_HIP(hipMalloc(&out, (size_t)GEMM_MAX_KS * rows * N * 4));
        QTTS_HIP(hipMalloc(&act, (size_t)rows * N * 2));
        QTTS_HIP(hipMalloc(&xb, (size_t)rows * N * 2 * 2));
        QTTS_HIP(hipMalloc(&x, (size_t)rows * N * 4));
        QTTS_HIP(hipMalloc(&ss, (size_t)(N / 16 + 1) * rows * 4));
        QTTS_HIP(hipMalloc(&nw, (size_t)N * 4));
        QTTS_HIP(hipMemset(W, 0x11, wbytes * copies));   // small finite bf16 values
        QTTS_HIP(hipMemset(X, 0x11, (size_t)rows * K * 2 * 2));
        QTTS_HIP(hipMemset(x, 0, (size_t)rows * N * 4));
        QTTS_HIP(hipMemset(nw, 0, (size_t)N * 4));
        const long x_lo = lo ? (long)rows * K : 0;
        hipStream_t st = nullptr;
        auto launch = [&](int i) {
            const WMat w((const char*)W + wbytes * (size_t)(i % copies), true);
            FusedEpi e{};
            if (kind == "k_split") {
                e.mode = FUSED_RESID; e.ss_ld = rows; e.resid = x; e.x_out = x; e.xb_out = xb; e.xb_lo = (long)rows * N; e.nw = nw; e.ss_out = ss;
                launch_gemm_k_split(st, w, X, K, x_lo, M, N, K, e, tuning_forced() == TUNE_SHARED);
#ifndef HARNESS_FUSED_ONLY
            } else if (kind == "slab") {
                launch_gemm(st, BF16, w, X, out, M, N, K, gemm_pick_ks(M, N, K, BF16), K, x_lo);
#endif
            } else {
                e.mode = mode ? FUSED_SWIGLU : FUSED_PLAIN; e.out_f32 = out; e.ld_out = N; e.act = act; e.act_lo = 0;
                if (kind == "stream") launch_gemm_stream(st, w, X, K, x_lo, M, N, K, e);
                else launch_gemm_full_k(st, w, X, K, x_lo, M, N, K, e);
            }
        };
        for (int i = 0; i < 10; ++i) launch(i);
        QTTS_HIP(hipDeviceSynchronize());
        hipEvent_t a, b;
        QTTS_HIP(hipEventCreate(&a));
        QTTS_HIP(hipEventCreate(&b));
        QTTS_HIP(hipEventRecord(a, st));
        for (int i = 0; i < iters; ++i) launch(i);
        QTTS_HIP(hipEventRecord(b, st));
        QTTS_HIP(hipEventSynchronize(b));
        float ms = 0;
        QTTS_HIP(hipEventElapsedTime(&ms, a, b));
        const double us = ms * 1e3 / iters;
        printf("%s M=%d N=%d K=%d lo=%d mode=%d: %.2f us/launch, %.1f GB/s of weights\n", kind.c_str(), M, N, K, (int)lo, mode, us, wbytes / us / 1e3);
        unsigned long long ts[2][16];
        QTTS_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_kernel_ts), sizeof(ts)));
        for (int b = 0; b < 2; ++b) {
            printf("  %s block:", b ? "last " : "first");
            for (int i = 1; i <= 6; ++i) printf(" t%d %+.2f", i, (double)(long long)(ts[b][i] - ts[0][0]) / 100.0);
            printf("   (own start %+.2f)\n", (double)(long long)(ts[b][0] - ts[0][0]) / 100.0);
        }
    } catch (const Error& err) {
        fprintf(stderr, "error %d: %s\n", err.code, err.what());
        return 1;
    }
    return 0;
}
