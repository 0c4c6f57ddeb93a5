// Developer scratch: what a wave-tiled MFMA loop can reach on this GPU.  512 workgroups of 4 waves (two per CU), every wave
// keeps 32 accumulator tiles (8 row fragments x 4 column tiles, as conv_tile_kernel) and runs v_mfma_f32_16x16x32_bf16 on them:
//   mode 0: MFMAs only (operands stay in registers)
//   mode 1: one ds_read_b128 per four MFMAs (the B fragment halves of the conv kernels), conflict-free addresses
//   mode 2: as 1 with the two-way conflicted addresses of the plain XOR swizzle
// Prints TFLOP/s.  hipcc --offload-arch=gfx950 -O3 mfmabench.hip -o mfmabench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters) {
    extern __shared__ uint4 tile[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = make_uint4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    f32x4_t acc[8][4];
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t a[4][2];
    for (int j = 0; j < 4; ++j)
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) a[j][h][e] = (__bf16)(float)(lane + j + h + e);
    const int r = lane & 15, g = lane >> 4;
    int q = (lane + 4) & 15;
    const int fr = MODE == 2 ? r : (((q & 7) << 1) | (q >> 3));
    auto pcol = [&](int row, int col) {
        if (MODE == 2) return col ^ (row & 7);
        return (((col >> 1) + (row >> 1)) & 7) << 1 | ((col ^ row) & 1);
    };
    for (int it = 0; it < iters; ++it) {
        const int tap = it & 3;
        const int xr = (fr + tap) * 16 + pcol(fr + tap, 2 * g);
        if (MODE == 0) {
            bf16x8_t b;
            for (int e = 0; e < 8; ++e) b[e] = (__bf16)(float)(it + e);
#pragma unroll
            for (int t = 0; t < 16; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t >> 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][t & 1], b, acc[t >> 1][j], 0, 0, 0);
        } else {
            uint4 hq[16];
#pragma unroll
            for (int t = 0; t < 3; ++t) hq[t] = tile[(16 * (t >> 1) * 16 + xr) ^ (t & 1)];
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                if (t + 3 < 16) {
                    hq[t + 3] = tile[(16 * ((t + 3) >> 1) * 16 + xr) ^ ((t + 3) & 1)];
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                bf16x8_t b;
                __builtin_memcpy(&b, &hq[t], 16);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t >> 1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j][t & 1], b, acc[t >> 1][j], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, int blocks) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 65536, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 64 * 16384.0;
    printf("mode %d blocks %d: %.3f ms  %.1f TFLOP/s\n", MODE, blocks, ms, flops / ms / 1e9);
}

int main() {
    float* out;
    hipMalloc(&out, 2048 * 256 * 4);
    for (int blocks : {256, 512}) {
        run<0>(out, blocks);
        run<1>(out, blocks);
        run<2>(out, blocks);
    }
    return 0;
}
