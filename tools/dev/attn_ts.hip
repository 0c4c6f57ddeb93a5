// In-kernel timeline of the decode attention kernel (development probe): compiles the kernel source with the wall-clock marks
// switched on, launches it back to back on a code-predictor-like or talker-like shape and prints where the first and the last
// block of a launch spend their time, next to the average launch-to-launch time.
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -I include -I qwen3-tts-rs_amd/csrc -o tools/dev/attn_ts tools/dev/attn_ts.hip
//   tools/dev/attn_ts M ctx nh nkv
#define QTTS_KERNEL_TS 1
#include "../../qwen3-tts-rs_amd/csrc/k_attn.hip"

#include <cstdio>
#include <vector>

using namespace qtts;

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 64, ctx = argc > 2 ? atoi(argv[2]) : 9, nh = argc > 3 ? atoi(argv[3]) : 16,
              nkv = argc > 4 ? atoi(argv[4]) : 8, d = 128, iters = 200;
    const int uni = argc > 5 ? atoi(argv[5]) : 0;
    const int page_tokens = 64, max_pages = (ctx + page_tokens - 1) / page_tokens;
    const int ncol = (nh + 2 * nkv) * d;
    const size_t layer_elems = (size_t)M * max_pages * 2 * nkv * page_tokens * d;
    void *pool, *out;
    float *part, *qn, *cs;
    Tok* toks;
    hipMalloc(&pool, layer_elems * 2);
    hipMemset(pool, 0x11, layer_elems * 2);
    hipMalloc(&out, (size_t)M * nh * d * 2 * 2);
    hipMalloc(&part, (size_t)M * ncol * 4);
    hipMemset(part, 0, (size_t)M * ncol * 4);
    hipMalloc(&qn, d * 4);
    hipMemset(qn, 0, d * 4);
    hipMalloc(&cs, (size_t)(ctx + 1) * d * 4);
    hipMemset(cs, 0, (size_t)(ctx + 1) * d * 4);
    std::vector<Tok> h(M);
    for (int m = 0; m < M; ++m) h[m] = Tok{m, ctx - 1, 0, ctx};
    hipMalloc(&toks, M * sizeof(Tok));
    hipMemcpy(toks, h.data(), M * sizeof(Tok), hipMemcpyHostToDevice);
    QkvPostArgs a{};
    a.part = part; a.KS = 1; a.ks_stride = 0; a.bias = nullptr; a.q_norm = qn; a.k_norm = qn; a.eps = 1e-6f; a.cos = cs; a.sin = cs + (size_t)(ctx + 1) * d / 2;
    a.rope_mode = 0; a.toks = toks; a.M = M; a.nh = nh; a.nkv = nkv; a.d = d; a.layer = 0; a.q_out = nullptr; a.uni_len = uni ? ctx : 0;
    KVView kv{pool, layer_elems, page_tokens, nkv, d, max_pages, nullptr};
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 10; ++i) launch_attention_decode(st, BF16, a, kv, 0.088f, out, (long)M * nh * d);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_attention_decode(st, BF16, a, kv, 0.088f, out, (long)M * nh * d);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ts[2][16];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_kernel_ts), sizeof(ts));
    printf("M %d ctx %d nh %d nkv %d: %.2f us per launch\n", M, ctx, nh, nkv, ms * 1000 / iters);
    for (int b = 0; b < 2; ++b) {
        printf("  %s block:", b ? "last " : "first");
        for (int i = 1; i <= 6; ++i) printf(" t%d %+.2f", i, (double)(ts[b][i] - ts[0][0]) / 100.0);
        printf("   (own start %+.2f us after the first block's)\n", (double)(ts[b][0] - ts[0][0]) / 100.0);
    }
    return 0;
}
