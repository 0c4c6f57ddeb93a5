// Diagnostics only: pure HBM read streams with the access shapes the GEMM kernels use, to measure what a
// kernel of a given size can reach at best (launch ramp, tail and access-pattern effects included).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// pattern 0: each wave instruction reads 1 KiB contiguous (lane * 16 B)
// pattern 1: each wave instruction reads 16 rows x 64 B (row stride `ld` bytes), two instructions per 128-B span
template <int PATTERN, int UNROLL>
__global__ __launch_bounds__(256) void stream_kernel(const uint4* __restrict__ src, unsigned* sink, long n_pieces, int ld16) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long wid = (long)blockIdx.x * 4 + wave, nw = (long)gridDim.x * 4;
    uint4 acc = make_uint4(0, 0, 0, 0);
    if (PATTERN == 0) {
        // piece = UNROLL KiB contiguous; pieces are dealt round-robin to waves (every index < n_pieces * 64 * UNROLL)
        for (long p = wid; p < n_pieces; p += nw) {
            const uint4* base = src + p * (64 * UNROLL);
            uint4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = base[u * 64 + lane];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    } else {
        // piece = 16 rows x (UNROLL * 128) bytes: one unrolled burst of 16-row x 64-byte wave loads
        const int r = lane & 15, g = lane >> 4;
        const int segs = ld16 / (8 * UNROLL);
        for (long p = wid; p < n_pieces; p += nw) {
            const long rg = p / segs;
            const int c = (int)(p % segs) * 8 * UNROLL;
            const uint4* rowp = src + (rg * 16 + r) * ld16 + c;
            uint4 v[2 * UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) {
                v[2 * u] = rowp[u * 8 + g * 2];
                v[2 * u + 1] = rowp[u * 8 + g * 2 + 1];
            }
#pragma unroll
            for (int u = 0; u < 2 * UNROLL; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

void launch_stream(hipStream_t st, const void* src, unsigned* sink, long bytes, int blocks, int pattern, int unroll, int ld_bytes) {
    const int U = unroll >= 8 ? 8 : 4;
    QTTS_CHECK(blocks >= 1 && blocks <= 65535, ERR_SHAPE, "stream: bad block count");
    long pieces;
    if (pattern == 0) {
        QTTS_CHECK(bytes % (1024L * U) == 0, ERR_SHAPE, "stream: bytes must be a multiple of the unrolled KiB piece");
        pieces = bytes / (1024L * U);
    } else {
        QTTS_CHECK(ld_bytes % (128 * U) == 0 && bytes % ((long)ld_bytes * 16) == 0, ERR_SHAPE,
                   "stream: row length must be a multiple of the unrolled 128-byte step and rows a multiple of 16");
        pieces = bytes / ((long)ld_bytes * 16) * (ld_bytes / (128 * U));
    }
    const int ld16 = ld_bytes / 16;
    if (pattern == 0) {
        if (U == 8) hipLaunchKernelGGL((stream_kernel<0, 8>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
        else hipLaunchKernelGGL((stream_kernel<0, 4>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
    } else {
        if (U == 8) hipLaunchKernelGGL((stream_kernel<1, 8>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
        else hipLaunchKernelGGL((stream_kernel<1, 4>), dim3(blocks), dim3(256), 0, st, (const uint4*)src, sink, pieces, ld16);
    }
}

}  // namespace qtts
