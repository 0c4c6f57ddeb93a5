// One-launch code-predictor pass (k_cp_pass.hip): arguments and scratch layout.
#pragma once
#include "device_utils.hpp"
#include "kernels.hpp"

namespace qtts {

// the stack shape the kernel's work split is written for (the code predictor of the 0.6B / 1.7B checkpoints:
// qwen_tts/src/config/code_predictor_config.rs:96-140) - one kv head per XCD, 384 MLP channels per XCD
constexpr int CP_PASS_H = 1024, CP_PASS_I = 3072, CP_PASS_NH = 16, CP_PASS_NKV = 8, CP_PASS_D = 128;
constexpr int CP_PASS_ROWS = 64;        // token rows of a launch (fewer live rows: the rest compute on whatever the buffers hold)
constexpr int CP_PASS_MAX_LAYERS = 8;
constexpr size_t CP_PASS_FLAG_BYTES = 2 * 256 * 64 + 256;

struct CpPassLayer {
    const bf16_t *qkv, *o, *gate_up, *down;  // fragment-order bf16 (WMat packed); gate_up with interleaved rows (LayerW::gate_up_il)
    const float *qkv_bias, *o_bias, *q_norm, *k_norm, *ln1, *ln2;
};

struct CpPassArgs {
    CpPassLayer layer[CP_PASS_MAX_LAYERS];
    int layers;
    const float* final_norm;
    const bf16_t* head;      // [V][H] fragment order
    int V;
    float eps, scale;
    const float *cos, *sin;  // RoPE table rows of position `pos` (d / 2 values each)
    int M, pos, ctx;         // live rows; every row sits at position pos and attends keys 0 .. ctx - 1 (ctx = pos + 1)
    const float* x_in;       // [64][H] f32: residual rows entering layer 0
    const float* qkv0;       // [64][(nh + 2 nkv) d] f32: q | k | v rows of layer 0
    KVView kv;
    float* logits;           // [64][ld_logits] f32
    int ld_logits;
    // scratch (cp_pass_bind_scratch): must be zero before the first launch and after a launch that raised `err`
    unsigned *lflags, *gflags, *err;
    void *xn, *attn, *act;
    float *qkv, *slab;
    unsigned spin_limit;
    unsigned long long* ts;  // development aid: 2 x 128 wall-clock marks (100 MHz) of the first and the last workgroup, or null
};

size_t cp_pass_scratch_bytes();
void cp_pass_bind_scratch(CpPassArgs& a, void* base);
// Launches 256 workgroups that wait for each other: the launch needs the whole device's CUs to become available to it at some
// point (other kernels may run beside it, as long as they end on their own).
void launch_cp_pass(hipStream_t st, const CpPassArgs& a);

}  // namespace qtts
