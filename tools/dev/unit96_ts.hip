// In-kernel timeline of the 96-channel fused unit (development probe): k_conv.hip compiled with its wall-clock marks; one unit
// over M rows (one strip) with constant weights; prints where a steady-state tile of the first and the last workgroup spent
// its time.   unit96_ts M dil
// Build: hipcc -std=c++17 -O3 --offload-arch=gfx950 -Dconv_unit96_kernel=conv_unit96_kernel_ts -I include -I qwen3-tts-rs_amd/csrc -o tools/dev/unit96_ts tools/dev/unit96_ts.hip -L qwen3-tts-rs_amd -lqtts_hip
#include <cstdio>
#include <cstdlib>
#include <vector>

#define QTTS_KERNEL_TS 1
#include "../../qwen3-tts-rs_amd/csrc/k_conv.hip"

using namespace qtts;

int main(int argc, char** argv) {
    const long M = argc > 1 ? atol(argv[1]) : 13670481;
    const int dil = argc > 2 ? atoi(argv[2]) : 1;
    const int C = 96, P = 64;
    try {
        void *W1, *W2;
        float *rin, *rout, *par;
        int* lens;
        const size_t rows = (size_t)M + P + CONV_HALO_SLACK + 128;
        QTTS_HIP(hipMalloc(&W1, (size_t)6 * 21 * 64 * 16));
        QTTS_HIP(hipMalloc(&W2, (size_t)6 * 3 * 64 * 16));
        QTTS_HIP(hipMalloc(&rin, rows * C * 4));
        QTTS_HIP(hipMalloc(&rout, rows * C * 4));
        QTTS_HIP(hipMalloc(&par, 8 * C * 4));
        QTTS_HIP(hipMalloc(&lens, 4));
        QTTS_HIP(hipMemset(W1, 0x11, (size_t)6 * 21 * 64 * 16));
        QTTS_HIP(hipMemset(W2, 0x11, (size_t)6 * 3 * 64 * 16));
        QTTS_HIP(hipMemset(rin, 0, rows * C * 4));
        std::vector<float> ones(8 * C, 1.0f);
        QTTS_HIP(hipMemcpy(par, ones.data(), 8 * C * 4, hipMemcpyHostToDevice));
        const int len = (int)M;
        QTTS_HIP(hipMemcpy(lens, &len, 4, hipMemcpyHostToDevice));
        ConvEpi e{};
        e.M = M; e.N = C; e.r = 1; e.Cout = C; e.bias = par; e.resid_f32 = rin; e.lens = lens; e.S_in = M + P; e.S_out = M + P; e.P_out = P; e.ld_out = C;
        e.out_f32 = rout;
        hipStream_t st = nullptr;
        auto launch = [&]() {
            if (!launch_conv_unit96(st, W1, W2, dil, par + C, par + 2 * C, par + 3 * C, e, rin + (size_t)(P - 6 * dil) * C, par + 4 * C, par + 5 * C)) {
                fprintf(stderr, "shape not covered\n");
                exit(1);
            }
        };
        launch();
        QTTS_HIP(hipDeviceSynchronize());
        hipEvent_t a, b;
        QTTS_HIP(hipEventCreate(&a));
        QTTS_HIP(hipEventCreate(&b));
        QTTS_HIP(hipEventRecord(a, st));
        for (int i = 0; i < 5; ++i) launch();
        QTTS_HIP(hipEventRecord(b, st));
        QTTS_HIP(hipEventSynchronize(b));
        float ms = 0;
        QTTS_HIP(hipEventElapsedTime(&ms, a, b));
        ms /= 5;
        printf("unit96 M=%ld dil=%d: %.3f ms (%.2f us per tile and workgroup at 512 workgroups)\n", M, dil, ms, ms * 1e3 / ((M + 63) / 64 / 512.0));
        unsigned long long ts[2][16];
        QTTS_HIP(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_kernel_ts), sizeof(ts)));
        const char* names[] = {"", "rows in LDS", "K walk", "tail + barrier", "1x1", "staging + barrier", "epilogue"};
        for (int bk = 0; bk < 2; ++bk) {
            printf("  %s workgroup, fourth tile:", bk ? "last " : "first");
            for (int i = 1; i <= 6; ++i) printf(" %s %+.2f", names[i], (double)(long long)(ts[bk][i] - ts[bk][0]) / 100.0);
            printf("\n");
        }
    } catch (const Error& err) {
        fprintf(stderr, "error %d: %s\n", err.code, err.what());
        return 1;
    }
    return 0;
}
