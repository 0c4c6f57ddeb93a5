// Bit-for-bit check of the cross-lane reductions built from DPP row rotations and the gfx950 permlane swaps against the
// __shfl_xor butterflies they replace (development probe).   hipcc -O3 --offload-arch=gfx950 -o tools/dev/dppcheck tools/dev/dppcheck.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__device__ __forceinline__ float ref_sum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float ref_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// The swaps work in place on two registers.  With both holding v, the first result is (v.lo, v.lo) in the 32-lane halves (16-lane
// rows: (r0, r0, r2, r2)) and the second (v.hi, v.hi) ((r1, r1, r3, r3)): combining the two gives every lane "own op partner".
template <typename F>
__device__ __forceinline__ float fold32(float v, F op) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return op(a, b);
}
template <typename F>
__device__ __forceinline__ float fold16(float v, F op) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return op(a, b);
}
__device__ __forceinline__ float new_sum(float v) {
    auto add = [](float x, float y) { return x + y; };
    v = fold32(v, add);
    v = fold16(v, add);
    v += dpp<0x128>(v);
    v += dpp<0x124>(v);
    v += dpp<0x122>(v);
    v += dpp<0x121>(v);
    return v;
}
__device__ __forceinline__ float new_max(float v) {
    auto mx = [](float x, float y) { return fmaxf(x, y); };
    v = fold32(v, mx);
    v = fold16(v, mx);
    v = fmaxf(v, dpp<0x128>(v));
    v = fmaxf(v, dpp<0x124>(v));
    v = fmaxf(v, dpp<0x122>(v));
    v = fmaxf(v, dpp<0x121>(v));
    return v;
}
__global__ void k(const float* in, float* out) {
    const float v = in[blockIdx.x * 64 + threadIdx.x];
    float* o = out + (blockIdx.x * 64 + threadIdx.x) * 8;
    o[0] = ref_sum(v); o[1] = new_sum(v); o[2] = ref_max(v); o[3] = new_max(v);
    float s = v, t = v;
    for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    t += dpp<0x128>(t); t += dpp<0x124>(t); t += dpp<0x122>(t); t += dpp<0x121>(t);
    o[4] = s; o[5] = t;
    o[6] = v + __shfl_xor(v, 16, 64); o[7] = fold16(v, [](float x, float y) { return x + y; });
}
int main() {
    const int B = 256, n = B * 64;
    float *in, *out;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 32);
    float* h = (float*)malloc(n * 4); float* ho = (float*)malloc(n * 32);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * expf((rand() % 20) - 10.f);
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(B), dim3(64), 0, 0, in, out);
    hipMemcpy(ho, out, n * 32, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; ++i)
        for (int p = 0; p < 8; p += 2)
            if (memcmp(&ho[i * 8 + p], &ho[i * 8 + p + 1], 4)) { if (bad < 5) printf("mismatch lane %d pair %d: %g %g\n", i, p, ho[i*8+p], ho[i*8+p+1]); ++bad; }
    printf("in[0] %g in[16] %g in[32] %g  fold16: ref %g new %g\n", h[0], h[16], h[32], ho[6], ho[7]);
    printf("%s (%d mismatches)\n", bad ? "FAIL" : "OK", bad);
    return bad != 0;
}
