// Host-side model state: configs parsed from the reference's JSON files, device-resident weights and
// the per-GPU context.  Mirrors the data the reference keeps in ConditionalGeneration / TokenizerV2
// (qwen_tts/src/nn/generation.rs:73-78, audio/tokenizer/v2/mod.rs:36-40).
#pragma once
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "common.hpp"
#include "json_min.hpp"
#include "kernels.hpp"

namespace qtts {

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
};

struct DevBuf {  // owning device allocation
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) { o.p = nullptr; o.bytes = 0; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) { release(); p = o.p; bytes = o.bytes; o.p = nullptr; o.bytes = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    void alloc(size_t n) {
        release();
        if (n == 0) n = 16;
        QTTS_HIP(hipMalloc(&p, n));
        bytes = n;
    }
    void ensure(size_t n) {
        if (n > bytes) alloc(n);
    }
    template <typename T>
    T* as() const { return (T*)p; }
};

struct Tensor {
    DevBuf buf;
    int dtype = F32;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct StackCfg {  // one transformer stack (talker or code predictor)
    int hidden = 0, inter = 0, layers = 0, nh = 0, nkv = 0, d = 0;
    float eps = 1e-6f;
    double theta = 1e4;
    int window = 0;  // 0 = full attention
    bool bias = false;
};

struct ModelCfg {
    StackCfg talker, cp;
    int vocab = 3072, text_hidden = 2048, text_vocab = 151936, groups = 32, cp_vocab = 2048;
    int eos = 4198, bos = 4197, pad = 4196, think = 4202, nothink = 4203, think_bos = 4204, think_eos = 4205;
    int tts_bos = 151672, tts_eos = 151673, tts_pad = 151671;
    std::map<std::string, int> spk_id, language_id;
    std::map<std::string, std::string> spk_dialect;
    bool has_speaker_encoder = false;
    // speaker encoder (config/speaker_encoder_config.rs)
    int mel_dim = 128, enc_dim = 1024, att_channels = 128, res2net_scale = 8, se_channels = 128;
    std::vector<int> enc_channels{512, 512, 512, 512, 1536}, enc_kernels{5, 3, 3, 3, 1}, enc_dilations{1, 2, 3, 4, 1};
};

struct CodecCfg {  // audio/decoder/config.rs
    bool present = false;
    int codebook_size = 2048, hidden = 1024, latent = 1024, codebook_dim = 256;
    double rope_theta = 10000.0;
    int heads = 16, kv_heads = 16, head_dim = 64;
    bool attention_bias = false;
    int window = 72, inter = 3072;
    float eps = 1e-5f;
    int layers = 8, num_quantizers = 16;
    std::vector<int> upsample_rates{8, 5, 4, 3}, upsampling_ratios{2, 2};
    int decoder_dim = 1536;
    int total_upsample() const {
        int t = 1;
        for (int r : upsample_rates) t *= r;
        for (int r : upsampling_ratios) t *= r;
        return t;
    }
};

struct LayerW {
    WMat qkv, o, gate_up, down;  // model dtype
    WMat gate_up_il;  // gate|up rows interleaved in groups of four (SwiGLU inside the GEMM), fused decode stacks only
    const float *qkv_bias = nullptr, *o_bias = nullptr, *q_norm = nullptr, *k_norm = nullptr, *ln1 = nullptr, *ln2 = nullptr;
    const float *ls_attn = nullptr, *ls_mlp = nullptr;  // LayerScale (codec transformer only)
};

struct StackW {
    StackCfg cfg;
    std::vector<LayerW> layers;
    const float* final_norm = nullptr;
    DevBuf cos, sin;  // [max_pos][d/2] f32
    int max_pos = 0;
};

struct Model;
struct GenState;    // generate.cpp
struct CodecState;    // codec.cpp: the decoder's weights (shared by the sessions of a model)
struct CodecScratch;  // codec.cpp: its run-time buffers (one set per model handle, session or not)
struct SpeakerState;
struct EncoderState;  // encoder.cpp

// The speaker encoder and the codec encoder keep their weights and their workspaces in one state object each; sessions of one
// model share those objects, so calls into them are serialised (a call returns only when its result is on the host, so the
// workspaces are free again when the lock is released).  The codec decoder's buffers belong to the handle (CodecScratch).
struct SharedLocks {
    std::mutex speaker, encoder;
};

// Code-predictor passes 1 .. groups - 2 start from the embedding row of the code just drawn.  What the first layer makes of such a
// row before attention - the projected residual row x and its q|k|v row - depends on (codebook, code) only, so both are tabulated
// once per model with the decode kernels themselves (same values bit for bit) and a pass gathers two rows instead of running a
// projection and a q|k|v GEMM.  Shared by the sessions of a model.
struct CpMemo {
    std::mutex mu;
    bool ready = false;
    int tables = 0, Vc = 0, Hc = 0, nqkv = 0;
    DevBuf x, qkv;  // [tables][Vc][Hc] and [tables][Vc][nqkv] f32
};

// Decode kernel choice of a weight set (kernels.hpp, TUNE_*): shared by a model and its sessions, set once.
struct Tuning {
    std::mutex mu;
    int mode = TUNE_UNSET;
    // the choice in force; `want` (TUNE_ALONE / TUNE_SHARED) becomes the choice if none has been made yet
    int latch(int want) {
        std::lock_guard<std::mutex> lock(mu);
        if (mode == TUNE_UNSET) mode = tuning_forced() != TUNE_UNSET ? tuning_forced() : want;
        return mode;
    }
    int peek() {
        std::lock_guard<std::mutex> lock(mu);
        return mode;
    }
};

struct Model {
    Ctx* ctx = nullptr;
    bool is_view = false;  // a session: weights borrowed from another model, which must outlive it
    std::shared_ptr<SharedLocks> locks = std::make_shared<SharedLocks>();
    std::shared_ptr<CpMemo> cp_memo = std::make_shared<CpMemo>();
    std::shared_ptr<Tuning> tuning = std::make_shared<Tuning>();
    int dtype = BF16;
    ModelCfg cfg;
    CodecCfg codec;
    bool finalized = false;
    std::map<std::string, Tensor> raw;     // as loaded, by HF key
    std::vector<DevBuf> owned;             // fused / derived tensors
    StackW talker, cp;
    const void *codec_emb = nullptr, *text_emb = nullptr;
    WMat codec_head, fc1, fc2, proj;
    const float *fc1_b = nullptr, *fc2_b = nullptr, *proj_b = nullptr;
    std::vector<const void*> cp_emb;
    std::vector<WMat> cp_head;
    DevBuf cp_emb_ptrs;                    // device array of G-1 table pointers
    std::shared_ptr<GenState> gen;
    std::shared_ptr<CodecState> codec_state;
    std::shared_ptr<CodecScratch> codec_scratch;  // created by the handle's first codec call
    std::shared_ptr<SpeakerState> spk_state;
    std::shared_ptr<EncoderState> enc_state;
    double timing[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<double> gemm_profile;      // rows of (M, N, K, launches, total us) from the last profiled generate

    Model();
    ~Model();
    void parse_config(const std::string& config_json, const char* codec_json);
    void load_tensor(const std::string& key, const void* data, int src_dtype, const int64_t* shape, int rank, bool on_device);
    void finalize();
    const Tensor& get(const std::string& key) const;
    bool has(const std::string& key) const { return raw.count(key) != 0; }
    const float* f32(const std::string& key) const;
    const void* mat(const std::string& key) const;
    void ensure_rope(StackW& s, int max_pos);
    // Makes this (empty) model a second decode stream of `base`: same weights (pointers), own RoPE tables, KV caches, workspaces
    // and frame graph, running on this->ctx.  `base` must be finalized.
    void init_view(const Model& base);
};

// Fuses q|k|v and gate|up of every layer under `prefix` and fills the per-layer weight table.
void build_stack(Model& m, StackW& s, const StackCfg& c, const std::string& prefix, bool qk_norm, bool layer_scale,
                 bool fused_layout = false);
// [N][K] matrix of the model dtype -> fragment order in a buffer the model owns (asynchronous on the model's stream; `src`
// must stay alive until the stream has been synchronised).  Row-major WMat of `src` itself when the shape does not tile.
WMat pack_matrix(Model& m, const void* src, int64_t N, int64_t K);
bool pack_weights_enabled();  // QTTS_PACK_WEIGHTS=0 keeps every GEMM weight row-major

// cos/sin tables on the host exactly as the reference computes them
void rope_table_default(int d, double theta, int max_pos, std::vector<float>& cos, std::vector<float>& sin);  // rope/core.rs:150-160
void rope_table_simple(int d, double theta, int max_pos, std::vector<float>& cos, std::vector<float>& sin);   // rope/simple.rs:30-64

}  // namespace qtts
