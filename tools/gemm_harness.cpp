// Stand-alone driver of the decode GEMM kernels (no Python, no torch): launches one kernel variant back to back on rotating
// weight copies and prints the average launch time and the weight bytes per second.  Meant for rocprofv3 (the program itself
// goes after `--`, so counter collection needs no interpreter in between) and for comparing kernel variants.
//   gemm_harness <stream|full_k|k_split|slab> M N K lo(0/1) [iters] [mode: 0 plain, 1 swiglu]
// Build: hipcc -std=c++17 -O3 --offload-arch=gfx950 -I include -I qwen3-tts-rs_amd/csrc tools/gemm_harness.cpp \
//        -L qwen3-tts-rs_amd -lqtts_hip -Wl,-rpath,'$ORIGIN/../qwen3-tts-rs_amd' -o tools/gemm_harness
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kernels.hpp"

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
            } else if (kind == "frag") {
                if (mode == 2) { e.mode = FUSED_RESID; e.ss_ld = rows; e.resid = x; e.x_out = x; e.xb_out = xb; e.xb_lo = 0; e.nw = nw; e.ss_out = ss; }
                else { e.mode = FUSED_PLAIN; e.out_f32 = out; e.ld_out = N; }
                if (!launch_gemm_frag(st, w, X, N, K, e)) throw Error(ERR_SHAPE, "frag: shape not covered");
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
        // HARNESS_STREAMS=n: the same chain of launches replayed from one graph per stream on n streams at once (all write the
        // same outputs: only the timing means anything)
        if (const char* hs = getenv("HARNESS_STREAMS")) {
            const int ns = atoi(hs);
            std::vector<hipStream_t> sts(ns);
            std::vector<hipGraphExec_t> ex(ns);
            for (int q = 0; q < ns; ++q) {
                QTTS_HIP(hipStreamCreateWithFlags(&sts[q], hipStreamNonBlocking));
                hipGraph_t gr;
                st = sts[q];
                QTTS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
                // HARNESS_ALIGNED=1: every stream walks the weight copies in the same order from the same start, so a stream that
                // runs just behind another finds its weights in the memory-side cache; default: seven copies apart
                static const bool aligned = getenv("HARNESS_ALIGNED") && atoi(getenv("HARNESS_ALIGNED")) != 0;
                for (int i = 0; i < iters; ++i) launch(i + (aligned ? 0 : q * 7));
                QTTS_HIP(hipStreamEndCapture(st, &gr));
                QTTS_HIP(hipGraphInstantiate(&ex[q], gr, nullptr, nullptr, 0));
                QTTS_HIP(hipGraphDestroy(gr));
            }
            for (int q = 0; q < ns; ++q) QTTS_HIP(hipGraphLaunch(ex[q], sts[q]));
            QTTS_HIP(hipDeviceSynchronize());
            const auto t0 = std::chrono::steady_clock::now();
            for (int rep = 0; rep < 5; ++rep)
                for (int q = 0; q < ns; ++q) QTTS_HIP(hipGraphLaunch(ex[q], sts[q]));
            QTTS_HIP(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / (5.0 * iters);
            printf("%s M=%d N=%d K=%d lo=%d mode=%d on %d streams: %.2f us per launch per stream, %.2f us per launch overall\n", kind.c_str(), M, N, K,
                   (int)lo, mode, ns, us, us / ns);
            return 0;
        }
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
    } catch (const Error& err) {
        fprintf(stderr, "error %d: %s\n", err.code, err.what());
        return 1;
    }
    return 0;
}
