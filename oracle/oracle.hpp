// CPU oracle for the Qwen3-TTS inference hot path -- TEST INFRASTRUCTURE ONLY.
//
// A plain C++17 / fp32 restatement of the algorithm the reference implements on top of
// Candle (tardigrade-dot/qwen3-tts-rs, crate qwen_tts).  Every function cites the reference
// file:line it follows.  Nothing under oracle/ is linked into or called by the shipped
// library; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it,
// and only as the checker / the reported CPU baseline.
//
// PARITY STATUS.  The reference cannot be built here (Rust, Candle not vendored) and its own tests pin only a handful of micro
// known-answer vectors (MT19937/Box-Muller seed 42, logit-processor toy vectors, rotate_half, reflect_pad, Hann end points,
// mask / position-id vectors, text normalisation); those are reproduced in tests/ against this oracle.  For the network
// outputs the oracle is pinned instead on independent public implementations of the same architectures, run in the build
// container on the same synthetic weights (generators and fixtures under tests/golden/): talker forward and the code-predictor
// loop on transformers' Qwen3Model, the codec decoder on its Qwen3-Omni Code2Wav modules, the codec encoder on its MimiModel,
// the speaker encoder on its ECAPA_TimeDelayNet (Qwen2.5-Omni), the mel front end on torch.stft.  What stays "parity unpinned"
// (this restatement's own values only): prompt assembly and the generate loop's EOS / suppress / min-new-token rules beyond
// the reference's toy vectors.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

struct Tensor {
    std::vector<int64_t> shape;
    std::vector<float> v;
    int64_t dim(int i) const { return shape[(size_t)i]; }
    const float* p() const { return v.data(); }
};
using Weights = std::unordered_map<std::string, Tensor>;

// ---------------------------------------------------------------------------------------
// Arithmetic primitives.  Summation order is fixed (8 strided partial sums, pairwise
// combine, scalar tail) so that results do not depend on compiler vectorisation choices.
// ---------------------------------------------------------------------------------------
inline float dot(const float* a, const float* b, int64_t n) {
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int64_t i = 0;
    for (; i + 8 <= n; i += 8)
        for (int j = 0; j < 8; ++j) s[j] += a[i + j] * b[i + j];
    float t = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    for (; i < n; ++i) t += a[i] * b[i];
    return t;
}

// y[m][n] = sum_k x[m][k] * w[n][k] (+ b[n]) -- candle_nn::Linear (weight is (out, in)).
void linear(const float* x, const float* w, const float* b, float* y, int64_t M, int64_t N, int64_t K);

// candle_nn::rms_norm CPU kernel: f32 sum of squares, m = sqrt(sum/n + eps), y = (x / m) * w.
// Reference call site: qwen_tts/src/nn/norm.rs:17-23.
void rmsnorm(const float* x, const float* w, float* y, int64_t rows, int64_t n, float eps);

// candle_nn::ops::softmax_last_dim on f32: max-subtract, exp, sequential sum, divide.
void softmax_inplace(float* x, int64_t n);

inline float silu(float v) { return v / (1.0f + std::exp(-v)); }

// ---------------------------------------------------------------------------------------
// RNG: MT19937 + the 24-bit uniform of qwen_tts/src/nn/mt_rng.rs:64-81 and the vectorised
// Box-Muller of :116-162 (rand_mt 4.2.2 `Mt::new(seed as u32)` == init_genrand).
// ---------------------------------------------------------------------------------------
struct MT19937 {
    uint32_t mt[624];
    int idx;
    explicit MT19937(uint32_t seed) { reseed(seed); }
    void reseed(uint32_t seed);
    uint32_t next_u32();
    // A strided view (stride n, offset b) yields row b's share of ONE stream that n rows draw from in turn: the reference
    // keeps a single global generator (mt_rng.rs:22) and, per frame, draws for rows 0 .. n-1 at the talker (generation.rs:
    // 1055-1060) and then again for rows 0 .. n-1 at every residual codebook (code_predictor.rs:468-472,517-522), so row b
    // owns the elements b, b + n, b + 2n, ... of the stream.  stride 1 / offset 0 is the plain stream.
    int stride = 1, offset = 0;
    bool primed = false;
    float uniform() {
        if (!primed) {
            for (int i = 0; i < offset; ++i) next_u32();
            primed = true;
        }
        const uint32_t v = next_u32();
        for (int i = 1; i < stride; ++i) next_u32();
        return (float)(v & 0x00FFFFFFu) / 16777216.0f;
    }
};
// sample_vectorized (mt_rng.rs:116-162) for count a multiple of 16.
void randn_vectorized(MT19937& rng, float* out, int count);

// ---------------------------------------------------------------------------------------
// Sampling -- qwen_tts/src/nn/sampling.rs
// ---------------------------------------------------------------------------------------
struct SamplingConfig {  // sampling.rs:13-64
    double temperature = 0.9;
    int top_k = 50;
    double top_p = 1.0;
    int do_sample = 1;
    double repetition_penalty = 1.05;
    int min_new_tokens = 2;
    int eos_token_id = -1;       // -1 = None
    int suppress_lo = 0, suppress_hi = 0;  // suppress ids in [lo, hi) except eos (generation.rs:848-853)
    int subtalker_do_sample = 1;
    double subtalker_temperature = 0.9;
    int subtalker_top_k = 50;
    double subtalker_top_p = 1.0;
    SamplingConfig for_subtalker() const;  // sampling.rs:109-125
};
// sample_token_with_step (sampling.rs:310-359) + sample_from_probs (:362-397).
// `u` supplies the one uniform the draw consumes (nullptr allowed when !do_sample).
// Sets *used_u to 1 when a uniform was consumed.
int64_t sample_token(const float* logits, int64_t V, const SamplingConfig& cfg,
                     const std::vector<int64_t>& history, MT19937* rng);
// Individual processors, exposed for the reference's own toy-vector tests (sampling.rs:435-601).
void apply_repetition_penalty(std::vector<float>& l, const std::vector<int64_t>& hist, double penalty);
void apply_temperature(std::vector<float>& l, double t);
void apply_top_k(std::vector<float>& l, int k);
void apply_top_p(std::vector<float>& l, double p);

// ---------------------------------------------------------------------------------------
// Transformer pieces shared by talker and code predictor.
// ---------------------------------------------------------------------------------------
struct AttnShape {
    int hidden = 0, nh = 0, nkv = 0, d = 0, inter = 0, layers = 0;
    float eps = 1e-6f;
    double theta = 1e6;
    int window = -1;  // sliding window (keys with 0 <= qpos - kpos < window), -1 = none
    bool bias = false;
};

struct KVCacheLayer {
    std::vector<float> k, v;  // [nkv][cap][d]
    int len = 0, cap = 0;
};

struct RopeTable {  // rope/core.rs:150-160 + rope/standard.rs:72-97 / rope/talker.rs:67-92
    int half = 0;
    std::vector<float> inv_freq;
    void init(int d, double theta);
    void cos_sin(int64_t pos, float* c, float* s) const;  // half values each
};

enum MaskMode { MASK_NONE = 0, MASK_CAUSAL = 1 };

// One pre-norm decoder layer over T new tokens at positions pos0..pos0+T-1 with KV cache
// (nn/decoder_layer.rs:63-86, nn/attention/unified.rs:179-243, nn/attention/mod.rs:359-472,
// nn/mlp.rs:59-82).  x is [T][hidden], updated in place.
void set_round_probe(int mask);  // diagnostic: see oracle_tts.cpp
void probe_round(float* x, size_t n, int bit);
void decoder_layer(const Weights& w, const std::string& prefix, const AttnShape& a, const RopeTable& rope,
                   float* x, int T, int pos0, MaskMode mask, KVCacheLayer& cache);

// ---------------------------------------------------------------------------------------
// Model
// ---------------------------------------------------------------------------------------
struct ModelConfig {
    AttnShape talker, cp;
    int vocab = 3072, text_hidden = 2048, text_vocab = 151936, groups = 16, cp_vocab = 2048;
    int eos = 2150, bos = 2149, pad = 2148;
    int think = 2154, nothink = 2155, think_bos = 2156, think_eos = 2157;
    int tts_bos = 151672, tts_eos = 151673, tts_pad = 151671;
};

struct GenRequest {
    std::vector<uint32_t> text_ids;
    std::vector<uint32_t> instruct_ids;
    std::vector<int> codec_prefill;     // resolved [think, think_bos, lang, think_eos] or [nothink, ...]
    int speaker_token = -1;             // codec_embedding row, -1 = none
    std::vector<float> speaker_embed;   // external x-vector (hidden), empty = none
    bool non_streaming = true;
    // ICL voice clone (generation.rs:292-416, 629-696)
    std::vector<uint32_t> ref_text_ids;
    std::vector<uint32_t> ref_codes;    // [ref_T][groups]
    int ref_T = 0;
};

struct GenOutput {
    std::vector<uint32_t> codes;  // [num_steps][groups]
    int num_steps = 0;
    int effective_len = 0;
    std::vector<float> logits_trace;  // [num_steps][vocab] talker logits before processors
    std::vector<float> hidden_trace;  // [num_steps][hidden] post-norm hidden fed to the code predictor
    float min_margin = 1e30f;         // smallest top1-top2 gap seen among greedy decisions
};

struct Model {
    ModelConfig cfg;
    Weights w;
    RopeTable rope_t, rope_c;
    void finalize();

    // talker.rs:345-348 embed_and_project_text for n ids -> [n][hidden]
    void embed_and_project_text(const uint32_t* ids, int n, float* out) const;
    // generation.rs:510-799.  Returns prompt [P][hidden]; trailing [Tt][hidden]; tts_pad [hidden].
    void build_prompt(const GenRequest& r, std::vector<float>& prompt, int& P, std::vector<float>& trailing,
                      int& Tt, std::vector<float>& tts_pad) const;
    // talker.rs:126-155 + :392-404: runs T tokens through the talker, returns post-norm hidden of every
    // token [T][hidden] (logits are computed by the caller for the rows it needs).
    void talker_forward(float* x, int T, int pos0, std::vector<KVCacheLayer>& cache, std::vector<float>& hidden) const;
    void talker_logits(const float* hidden_row, float* logits) const;
    // code_predictor.rs:433-531
    void code_predictor(const float* talker_hidden, int64_t cb0, const SamplingConfig& sub, MT19937* rng,
                        uint32_t* out_codes, float* min_margin, std::vector<float>* logits_dbg) const;
    // talker.rs:461-479
    void sum_code_embeddings(const uint32_t* codes, float* out) const;
    // generation.rs:815-1194 at batch 1
    void generate(const GenRequest& r, const SamplingConfig& s, int max_new_tokens, MT19937* rng, bool trace,
                  GenOutput& out) const;
};

const Tensor& W(const Weights& w, const std::string& key);
bool has(const Weights& w, const std::string& key);

}  // namespace orc
