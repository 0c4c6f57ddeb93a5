// In-kernel timeline of the 256-row conv kernel (development probe): k_conv.hip compiled with its wall-clock marks; runs one
// conv shape with a transposed-conv style epilogue (f32 stream + activated bf16 copy) and prints where the first and the last
// workgroup spent their time.   conv_ts M N K taps Cout [iters]
// Build: hipcc -std=c++17 -O3 --offload-arch=gfx950 -I include -I qwen3-tts-rs_amd/csrc -o tools/dev/conv_ts tools/dev/conv_ts.hip -L qwen3-tts-rs_amd -lqtts_hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#define QTTS_KERNEL_TS 1
#include "../../qwen3-tts-rs_amd/csrc/k_conv.hip"

using namespace qtts;

int main(int argc, char** argv) {
    if (argc < 6) {
        fprintf(stderr, "usage: %s M N K taps Cout [iters]\n", argv[0]);
        return 2;
    }
    const long M = atol(argv[1]);
    const int N = atoi(argv[2]), K = atoi(argv[3]), taps = atoi(argv[4]), Cout = atoi(argv[5]);
    const int iters = argc > 6 ? atoi(argv[6]) : 5;
    const int C = K / taps, r = N / Cout;
    try {
        void *W, *X, *out2;
        float *outf, *bias, *al, *dn;
        int* lens;
        const long rows_in = M + CONV_HALO_SLACK + 64;
        QTTS_HIP(hipMalloc(&W, (size_t)N * K * 2));
        QTTS_HIP(hipMalloc(&X, (size_t)rows_in * C * 2));
        QTTS_HIP(hipMalloc(&outf, (size_t)(M * r + 8) * Cout * 4));
        QTTS_HIP(hipMalloc(&out2, (size_t)(M * r + 8) * Cout * 2));
        QTTS_HIP(hipMalloc(&bias, Cout * 4));
        QTTS_HIP(hipMalloc(&al, Cout * 4));
        QTTS_HIP(hipMalloc(&dn, Cout * 4));
        QTTS_HIP(hipMalloc(&lens, 4));
        QTTS_HIP(hipMemset(W, 0x11, (size_t)N * K * 2));
        QTTS_HIP(hipMemset(X, 0x11, (size_t)rows_in * C * 2));
        QTTS_HIP(hipMemset(bias, 0, Cout * 4));
        std::vector<float> ones(Cout, 1.0f);
        QTTS_HIP(hipMemcpy(al, ones.data(), Cout * 4, hipMemcpyHostToDevice));
        QTTS_HIP(hipMemcpy(dn, ones.data(), Cout * 4, hipMemcpyHostToDevice));
        const int len = (int)M;
        QTTS_HIP(hipMemcpy(lens, &len, 4, hipMemcpyHostToDevice));
        ConvEpi e{};
        e.M = M; e.N = N; e.r = r; e.Cout = Cout; e.bias = bias; e.lens = lens; e.S_in = M; e.S_out = M * r; e.P_out = 0; e.ld_out = Cout;
        e.out_f32 = outf; e.out2 = out2; e.alpha_exp = al; e.beta_den = dn;
        hipStream_t st = nullptr;
        const WMat w(W, false);
        auto launch = [&]() {
            if (!launch_conv_halo(st, BF16, w, X, K, C, taps, (long)C, e, 0)) { fprintf(stderr, "shape not covered\n"); exit(1); }
        };
        launch();
        QTTS_HIP(hipDeviceSynchronize());
        hipEvent_t a, b;
        QTTS_HIP(hipEventCreate(&a));
        QTTS_HIP(hipEventCreate(&b));
        QTTS_HIP(hipEventRecord(a, st));
        for (int i = 0; i < iters; ++i) launch();
        QTTS_HIP(hipEventRecord(b, st));
        QTTS_HIP(hipEventSynchronize(b));
        float ms = 0;
        QTTS_HIP(hipEventElapsedTime(&ms, a, b));
        ms /= iters;
        printf("conv M=%ld N=%d K=%d taps=%d r=%d: %.3f ms, %.1f TFLOP/s\n", M, N, K, taps, r, ms, 2.0 * M * N * K / ms / 1e9);
        unsigned long long ts[2][16];
        QTTS_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_kernel_ts), sizeof(ts)));
        for (int bk = 0; bk < 2; ++bk) {
            printf("  %s block: start %+.2f us |", bk ? "last " : "first", (double)(long long)(ts[bk][0] - ts[0][0]) / 100.0);
            const char* names[] = {"", "phase0 loaded", "phase1 loaded", "K walk done", "first epilogue pass", "end"};
            for (int i = 1; i <= 5; ++i) printf(" %s %+.2f", names[i], (double)(long long)(ts[bk][i] - ts[bk][0]) / 100.0);
            printf("\n");
        }
    } catch (const Error& err) {
        fprintf(stderr, "error %d: %s\n", err.code, err.what());
        return 1;
    }
    return 0;
}
