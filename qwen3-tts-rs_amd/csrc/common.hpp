// Shared declarations for the gfx950 Qwen3-TTS hot path (host side + kernel launchers).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace qtts {

enum DType : int { F32 = 0, BF16 = 1 };
inline size_t dtype_size(int dt) { return dt == BF16 ? 2 : 4; }

struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};
// error classes returned through the C ABI
constexpr int ERR_BAD_ARG = -1, ERR_HIP = -2, ERR_OOM = -3, ERR_SHAPE = -4, ERR_STATE = -5, ERR_IO = -6;

#define QTTS_HIP(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            throw ::qtts::Error(e_ == hipErrorOutOfMemory ? ::qtts::ERR_OOM : ::qtts::ERR_HIP,           \
                                std::string(#expr) + ": " + hipGetErrorString(e_) + " (" + __FILE__ + ":" + \
                                    std::to_string(__LINE__) + ")");                                     \
    } while (0)

#define QTTS_CHECK(cond, code, msg)                          \
    do {                                                     \
        if (!(cond)) throw ::qtts::Error((code), (msg));     \
    } while (0)

// Blocking copy on `st`.  Run-time paths never use the legacy stream (plain hipMemcpy): it synchronises with every blocking
// stream of the process, which fails while another session captures its frame graph.
inline void copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t st) {
    QTTS_HIP(hipMemcpyAsync(dst, src, bytes, kind, st));
    QTTS_HIP(hipStreamSynchronize(st));
}

// HIP event owned by a scope: destroyed on every way out, including the exceptions QTTS_HIP / QTTS_CHECK throw.
struct ScopedEvent {
    hipEvent_t e = nullptr;
    ScopedEvent() { QTTS_HIP(hipEventCreate(&e)); }
    ~ScopedEvent() {
        if (e) (void)hipEventDestroy(e);
    }
    ScopedEvent(const ScopedEvent&) = delete;
    ScopedEvent& operator=(const ScopedEvent&) = delete;
    operator hipEvent_t() const { return e; }
};

inline uint16_t f32_to_bf16(float f) {  // round-to-nearest-even, NaN kept quiet
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

}  // namespace qtts
