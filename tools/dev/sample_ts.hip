// In-kernel timeline of the sampling kernel (development probe, see attn_ts.hip).
//   hipcc -std=c++17 -O3 --offload-arch=gfx950 -I include -I qwen3-tts-rs_amd/csrc -o tools/dev/sample_ts tools/dev/sample_ts.hip
//   tools/dev/sample_ts B V top_k top_p
#define QTTS_KERNEL_TS 1
#include "../../qwen3-tts-rs_amd/csrc/k_sample.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

using namespace qtts;

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, V = argc > 2 ? atoi(argv[2]) : 2048, top_k = argc > 3 ? atoi(argv[3]) : 50;
    const float top_p = argc > 4 ? (float)atof(argv[4]) : 1.0f;
    const int G = 16, max_steps = 8, H = 1024, iters = 200;
    std::vector<float> lg((size_t)B * V), un((size_t)B * max_steps * G);
    srand(3);
    for (auto& v : lg) v = (rand() / (float)RAND_MAX - 0.5f) * 12.f;
    for (auto& v : un) v = rand() / (float)RAND_MAX * 0.999f;
    float *dl, *du;
    int *step, *active;
    unsigned* codes;
    void *table, *dst;
    hipMalloc(&dl, lg.size() * 4); hipMemcpy(dl, lg.data(), lg.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&du, un.size() * 4); hipMemcpy(du, un.data(), un.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&step, B * 4); hipMemset(step, 0, B * 4);
    hipMalloc(&active, B * 4); hipMemset(active, 1, B * 4);
    hipMalloc(&codes, (size_t)B * max_steps * G * 4);
    hipMalloc(&table, (size_t)V * H * 2); hipMemset(table, 0x11, (size_t)V * H * 2);
    hipMalloc(&dst, (size_t)B * H * 2);
    SampleArgs a{};
    a.logits = dl; a.V = V; a.B = B; a.temperature = 1.0f / 0.9f; a.top_k = top_k; a.top_p = top_p; a.do_sample = 1; a.rep_penalty = 1.0f;
    a.hist_count = nullptr; a.suppress_lo = 0; a.suppress_hi = 0; a.eos_id = -1; a.min_new_tokens = 0; a.step = step; a.active = active;
    a.max_steps = max_steps; a.uniforms = du; a.codes_base = codes; a.row_stride = (long)max_steps * G; a.G = G; a.g = 3; a.eos_step = nullptr;
    a.next_table = table; a.next_dst = dst; a.next_H = H; a.next_table_bf16 = 1; a.next_dst_bf16 = 1;
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int i = 0; i < 10; ++i) launch_sample(st, a);
    hipStreamSynchronize(st);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < iters; ++i) launch_sample(st, a);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long ts[2][16];
    hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_kernel_ts), sizeof(ts));
    std::vector<unsigned> hc((size_t)B * max_steps * G);
    hipMemcpy(hc.data(), codes, hc.size() * 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    for (int b = 0; b < B; ++b) sum = sum * 31 + hc[(size_t)b * max_steps * G + 3];
    printf("B %d V %d top_k %d top_p %.2f: %.2f us per launch, token checksum %llu\n", B, V, top_k, top_p, ms * 1000 / iters, sum);
    for (int b = 0; b < 2; ++b) {
        printf("  %s block:", b ? "last " : "first");
        for (int i : {1, 7, 8, 9, 3, 4, 5, 6}) printf(" t%d %+.2f", i, (double)(ts[b][i] - ts[0][0]) / 100.0);
        printf("\n");
    }
    return 0;
}
