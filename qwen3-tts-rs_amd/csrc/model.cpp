// Model construction: config parsing, weight upload / conversion, fused weights, RoPE tables.
#include "model.hpp"

#include <cmath>
#include <cstring>

#include "generate.hpp"

namespace qtts {

Model::Model() = default;
Model::~Model() = default;

static StackCfg stack_from(const Json& j, int def_layers, int def_inter, int def_nkv) {
    StackCfg s;
    s.hidden = j.int_or("hidden_size", 1024);
    s.inter = j.int_or("intermediate_size", def_inter);
    s.layers = j.int_or("num_hidden_layers", def_layers);
    s.nh = j.int_or("num_attention_heads", 16);
    s.nkv = j.int_or("num_key_value_heads", def_nkv);
    s.d = j.int_or("head_dim", 128);
    s.eps = (float)j.num_or("rms_norm_eps", 1e-6);
    s.theta = j.num_or("rope_theta", 10000.0);
    s.bias = j.bool_or("attention_bias", false);
    return s;
}

void Model::parse_config(const std::string& config_json, const char* codec_json) {
    const Json root = JsonParser::parse(config_json);
    const Json& t = root.at("talker_config");
    // defaults: config/talker_config.rs:218-283, config/code_predictor_config.rs:96-140, config/tts_config.rs:54-68
    cfg.talker = stack_from(t, 20, 2048, 2);
    // the talker passes config.sliding_window to attention whenever it is non-null (talker_config.rs:360-362)
    cfg.talker.window = t.has("sliding_window") ? t.int_or("sliding_window", 0) : 0;
    const Json& c = t.at("code_predictor_config");
    cfg.cp = stack_from(c, 5, 3072, 8);
    cfg.cp.window = 0;
    cfg.vocab = t.int_or("vocab_size", 3072);
    cfg.text_hidden = t.int_or("text_hidden_size", 2048);
    cfg.text_vocab = t.int_or("text_vocab_size", 151936);
    cfg.groups = t.int_or("num_code_groups", 32);
    {
        // Layers of the code predictor may be typed "sliding_attention" (nn/attention/standard.rs:28-39; layer types as
        // code_predictor_config.rs:172-186 derives them).  Its context never exceeds num_code_groups + 1 positions, so a window at
        // least that long (the default is 4096, code_predictor_config.rs:135-137) cuts nothing and the layer is a full-attention
        // layer; a shorter window on some layers only would need a per-layer key range, which is not built.
        bool sliding = c.bool_or("use_sliding_window", false) && c.int_or("max_window_layers", 28) < cfg.cp.layers;
        if (c.has("layer_types"))
            for (auto& e : c.at("layer_types").arr) sliding = sliding || e.str == "sliding_attention";
        const int win = c.has("sliding_window") ? c.int_or("sliding_window", 0) : 4096;  // null: no window
        QTTS_CHECK(!sliding || win <= 0 || win >= cfg.groups + 1, ERR_BAD_ARG,
                   "per-layer sliding attention in the code predictor with a window shorter than its context is not supported");
    }
    cfg.cp_vocab = c.int_or("vocab_size", 2048);
    cfg.eos = t.int_or("codec_eos_token_id", 4198);
    cfg.bos = t.int_or("codec_bos_id", 4197);
    cfg.pad = t.int_or("codec_pad_id", 4196);
    cfg.think = t.int_or("codec_think_id", 4202);
    cfg.nothink = t.int_or("codec_nothink_id", 4203);
    cfg.think_bos = t.int_or("codec_think_bos_id", 4204);
    cfg.think_eos = t.int_or("codec_think_eos_id", 4205);
    cfg.tts_pad = root.int_or("tts_pad_token_id", 151671);
    cfg.tts_bos = root.int_or("tts_bos_token_id", 151672);
    cfg.tts_eos = root.int_or("tts_eos_token_id", 151673);
    if (t.has("spk_id"))
        for (auto& kv : t.at("spk_id").obj) cfg.spk_id[kv.first] = (int)kv.second.num;
    if (t.has("codec_language_id"))
        for (auto& kv : t.at("codec_language_id").obj) cfg.language_id[kv.first] = (int)kv.second.num;
    if (t.has("spk_is_dialect"))
        for (auto& kv : t.at("spk_is_dialect").obj)
            if (kv.second.kind == Json::Str) cfg.spk_dialect[kv.first] = kv.second.str;
    QTTS_CHECK(cfg.talker.nh % cfg.talker.nkv == 0 && cfg.cp.nh % cfg.cp.nkv == 0, ERR_SHAPE, "heads must be a multiple of kv heads");
    QTTS_CHECK(cfg.groups >= 2 && cfg.groups <= 64, ERR_SHAPE, "num_code_groups out of range");
    if (root.has("speaker_encoder_config")) {
        const Json& s = root.at("speaker_encoder_config");
        cfg.has_speaker_encoder = true;
        cfg.mel_dim = s.int_or("mel_dim", 128);
        cfg.enc_dim = s.int_or("enc_dim", 1024);
        cfg.enc_channels = s.ints_or("enc_channels", cfg.enc_channels);
        cfg.enc_kernels = s.ints_or("enc_kernel_sizes", cfg.enc_kernels);
        cfg.enc_dilations = s.ints_or("enc_dilations", cfg.enc_dilations);
        cfg.att_channels = s.int_or("enc_attention_channels", 128);
        cfg.res2net_scale = s.int_or("enc_res2net_scale", 8);
        cfg.se_channels = s.int_or("enc_se_channels", 128);
    }
    if (codec_json) {
        const Json cr = JsonParser::parse(codec_json);
        const Json& d = cr.at("decoder_config");
        codec.present = true;
        codec.codebook_size = d.int_or("codebook_size", 2048);
        codec.hidden = d.int_or("hidden_size", 1024);
        codec.latent = d.int_or("latent_dim", 1024);
        codec.codebook_dim = d.int_or("codebook_dim", 256);
        codec.rope_theta = d.num_or("rope_theta", 10000.0);
        codec.heads = d.int_or("num_attention_heads", 16);
        codec.kv_heads = d.int_or("num_key_value_heads", 16);
        codec.head_dim = d.has("head_dim") ? d.int_or("head_dim", 64) : codec.hidden / codec.heads;
        codec.attention_bias = d.bool_or("attention_bias", false);
        codec.window = d.int_or("sliding_window", 72);
        codec.inter = d.int_or("intermediate_size", 3072);
        codec.eps = (float)d.num_or("rms_norm_eps", 1e-5);
        codec.layers = d.int_or("num_hidden_layers", 8);
        codec.num_quantizers = d.int_or("num_quantizers", 16);
        codec.upsample_rates = d.ints_or("upsample_rates", codec.upsample_rates);
        codec.upsampling_ratios = d.ints_or("upsampling_ratios", codec.upsampling_ratios);
        codec.decoder_dim = d.int_or("decoder_dim", 1536);
    }
}

static bool key_is_f32(const std::string& key, int rank) {
    return rank <= 1 || key.find("codebook") != std::string::npos || key.find("dwconv") != std::string::npos;
}

void Model::load_tensor(const std::string& key, const void* data, int src_dtype, const int64_t* shape, int rank, bool on_device) {
    QTTS_CHECK(!finalized, ERR_STATE, "load_tensor after finalize");
    QTTS_CHECK(data != nullptr && rank >= 1 && rank <= 4, ERR_BAD_ARG, "load_tensor: bad tensor " + key);
    QTTS_CHECK(src_dtype == F32 || src_dtype == BF16, ERR_BAD_ARG, "load_tensor: source dtype must be f32 or bf16");
    Tensor t;
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) {
        QTTS_CHECK(shape[i] > 0, ERR_SHAPE, "load_tensor: empty dimension in " + key);
        t.shape.push_back(shape[i]);
        n *= shape[i];
    }
    t.dtype = key_is_f32(key, rank) ? F32 : dtype;
    t.buf.alloc((size_t)n * dtype_size(t.dtype));
    QTTS_HIP(hipSetDevice(ctx->device));
    if (on_device) {
        if (src_dtype == t.dtype) {
            QTTS_HIP(hipMemcpyAsync(t.buf.p, data, (size_t)n * dtype_size(t.dtype), hipMemcpyDeviceToDevice, ctx->stream));
        } else if (src_dtype == F32) {
            launch_cast_rows(ctx->stream, (const float*)data, t.buf.p, t.dtype, n);
        } else {
            throw Error(ERR_BAD_ARG, "load_tensor_device: bf16 source can only feed a bf16 tensor (" + key + ")");
        }
        QTTS_HIP(hipStreamSynchronize(ctx->stream));
    } else if (src_dtype == t.dtype) {
        QTTS_HIP(hipMemcpy(t.buf.p, data, (size_t)n * dtype_size(t.dtype), hipMemcpyHostToDevice));
    } else {
        // convert on the host in bounded pieces
        const int64_t piece = 1 << 22;
        std::vector<uint8_t> tmp((size_t)std::min(n, piece) * dtype_size(t.dtype));
        for (int64_t o = 0; o < n; o += piece) {
            const int64_t m = std::min(piece, n - o);
            if (src_dtype == F32) {  // -> bf16
                const float* s = (const float*)data + o;
                uint16_t* d = (uint16_t*)tmp.data();
                for (int64_t i = 0; i < m; ++i) d[i] = f32_to_bf16(s[i]);
            } else {  // bf16 -> f32
                const uint16_t* s = (const uint16_t*)data + o;
                float* d = (float*)tmp.data();
                for (int64_t i = 0; i < m; ++i) d[i] = bf16_to_f32(s[i]);
            }
            QTTS_HIP(hipMemcpy((uint8_t*)t.buf.p + (size_t)o * dtype_size(t.dtype), tmp.data(), (size_t)m * dtype_size(t.dtype),
                               hipMemcpyHostToDevice));
        }
    }
    raw[key] = std::move(t);
}

const Tensor& Model::get(const std::string& key) const {
    auto it = raw.find(key);
    QTTS_CHECK(it != raw.end(), ERR_STATE, "missing tensor " + key);
    return it->second;
}
const float* Model::f32(const std::string& key) const {
    const Tensor& t = get(key);
    QTTS_CHECK(t.dtype == F32, ERR_STATE, "tensor " + key + " is not f32");
    return t.buf.as<float>();
}
const void* Model::mat(const std::string& key) const {
    const Tensor& t = get(key);
    QTTS_CHECK(t.dtype == dtype, ERR_STATE, "tensor " + key + " is not in the model dtype");
    return t.buf.p;
}

static void expect_shape(const Tensor& t, const std::string& key, std::initializer_list<int64_t> want) {
    std::vector<int64_t> w(want);
    QTTS_CHECK(t.shape == w, ERR_SHAPE, "tensor " + key + " has the wrong shape");
}

void rope_table_default(int d, double theta, int max_pos, std::vector<float>& cos, std::vector<float>& sin) {
    const int half = d / 2;
    cos.resize((size_t)max_pos * half);
    sin.resize((size_t)max_pos * half);
    std::vector<float> inv((size_t)half);
    for (int i = 0; i < half; ++i) inv[(size_t)i] = 1.0f / (float)std::pow(theta, (double)i * 2.0 / (double)d);
    for (int p = 0; p < max_pos; ++p)
        for (int i = 0; i < half; ++i) {
            const float f = inv[(size_t)i] * (float)p;  // f32 product, then glibc cosf/sinf (rope/talker.rs:82-89)
            cos[(size_t)p * half + i] = std::cos(f);
            sin[(size_t)p * half + i] = std::sin(f);
        }
}
void rope_table_simple(int d, double theta, int max_pos, std::vector<float>& cos, std::vector<float>& sin) {
    const int half = d / 2;
    cos.resize((size_t)max_pos * half);
    sin.resize((size_t)max_pos * half);
    for (int p = 0; p < max_pos; ++p)
        for (int i = 0; i < half; ++i) {
            const float inv = 1.0f / std::pow((float)theta, (float)i / (float)half);
            const float f = (float)p * inv;
            cos[(size_t)p * half + i] = std::cos(f);
            sin[(size_t)p * half + i] = std::sin(f);
        }
}

void Model::ensure_rope(StackW& s, int max_pos) {
    if (max_pos <= s.max_pos) return;
    int n = 256;
    while (n < max_pos) n *= 2;
    std::vector<float> c, sn;
    rope_table_default(s.cfg.d, s.cfg.theta, n, c, sn);
    s.cos.alloc(c.size() * 4);
    s.sin.alloc(sn.size() * 4);
    copy_sync(s.cos.p, c.data(), c.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    copy_sync(s.sin.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, ctx->stream);
    s.max_pos = n;
}

// GEMM weights are re-laid into MFMA fragment order once, here (kernels.hpp WMat); QTTS_PACK_WEIGHTS=0 keeps them row-major.
bool pack_weights_enabled() {
    static const bool on = !(getenv("QTTS_PACK_WEIGHTS") && atoi(getenv("QTTS_PACK_WEIGHTS")) == 0);
    return on;
}
WMat pack_matrix(Model& m, const void* src, int64_t N, int64_t K) {
    const size_t es = dtype_size(m.dtype);
    if (!pack_weights_enabled() || N % 16 != 0 || (K * (int64_t)es) % 128 != 0) return WMat(src, false);
    DevBuf d;
    d.alloc((size_t)N * K * es);
    launch_pack_weight(m.ctx->stream, src, d.p, (int)N, (long)(K * es));
    WMat w(d.p, true);
    m.owned.push_back(std::move(d));
    return w;
}
// packs a matrix loaded under `key` and marks the row-major copy for release
static WMat pack_raw(Model& m, const std::string& key, std::vector<std::string>& release) {
    const Tensor& t = m.get(key);
    QTTS_CHECK(t.shape.size() == 2, ERR_SHAPE, ("weight " + key + " is not a matrix").c_str());
    const WMat w = pack_matrix(m, m.mat(key), t.shape[0], t.shape[1]);
    if (w.packed) release.push_back(key);
    return w;
}

void build_stack(Model& m, StackW& s, const StackCfg& c, const std::string& prefix, bool qk_norm, bool layer_scale, bool fused_layout) {
    s.cfg = c;
    s.layers.resize((size_t)c.layers);
    const size_t es = dtype_size(m.dtype);
    hipStream_t st = m.ctx->stream;
    std::vector<std::string> release;
    std::vector<DevBuf> staging;  // row-major fused matrices, dropped once packed
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = prefix + "layers." + std::to_string(l) + ".";
        LayerW& w = s.layers[(size_t)l];
        const int64_t nq = (int64_t)c.nh * c.d, nk = (int64_t)c.nkv * c.d;
        expect_shape(m.get(p + "self_attn.q_proj.weight"), p + "self_attn.q_proj.weight", {nq, c.hidden});
        expect_shape(m.get(p + "self_attn.k_proj.weight"), p + "self_attn.k_proj.weight", {nk, c.hidden});
        expect_shape(m.get(p + "self_attn.v_proj.weight"), p + "self_attn.v_proj.weight", {nk, c.hidden});
        expect_shape(m.get(p + "self_attn.o_proj.weight"), p + "self_attn.o_proj.weight", {c.hidden, nq});
        expect_shape(m.get(p + "mlp.gate_proj.weight"), p + "mlp.gate_proj.weight", {c.inter, c.hidden});
        expect_shape(m.get(p + "mlp.up_proj.weight"), p + "mlp.up_proj.weight", {c.inter, c.hidden});
        expect_shape(m.get(p + "mlp.down_proj.weight"), p + "mlp.down_proj.weight", {c.hidden, c.inter});
        // fused q|k|v (unified.rs:189-191 issues three Linear calls on the same input)
        DevBuf qkv;
        qkv.alloc((size_t)(nq + 2 * nk) * c.hidden * es);
        QTTS_HIP(hipMemcpyAsync(qkv.p, m.mat(p + "self_attn.q_proj.weight"), (size_t)nq * c.hidden * es, hipMemcpyDeviceToDevice, st));
        QTTS_HIP(hipMemcpyAsync((char*)qkv.p + (size_t)nq * c.hidden * es, m.mat(p + "self_attn.k_proj.weight"), (size_t)nk * c.hidden * es, hipMemcpyDeviceToDevice, st));
        QTTS_HIP(hipMemcpyAsync((char*)qkv.p + (size_t)(nq + nk) * c.hidden * es, m.mat(p + "self_attn.v_proj.weight"), (size_t)nk * c.hidden * es, hipMemcpyDeviceToDevice, st));
        // fused gate|up (mlp.rs:30-39)
        DevBuf gu;
        gu.alloc((size_t)2 * c.inter * c.hidden * es);
        QTTS_HIP(hipMemcpyAsync(gu.p, m.mat(p + "mlp.gate_proj.weight"), (size_t)c.inter * c.hidden * es, hipMemcpyDeviceToDevice, st));
        QTTS_HIP(hipMemcpyAsync((char*)gu.p + (size_t)c.inter * c.hidden * es, m.mat(p + "mlp.up_proj.weight"), (size_t)c.inter * c.hidden * es, hipMemcpyDeviceToDevice, st));
        if (fused_layout && fused_gemm_supported(m.dtype, c.hidden)) {
            // rows [4 gate, 4 up, 4 gate, 4 up, ...]: the lanes that hold a gate value and the matching up value of one
            // token are then 16 apart in the MFMA result of a 16-row tile (k_gemm_fused.hip)
            QTTS_CHECK(c.inter % 8 == 0, ERR_SHAPE, "intermediate size must be a multiple of 8");
            DevBuf il;
            il.alloc((size_t)2 * c.inter * c.hidden * es);
            const size_t grp = (size_t)4 * c.hidden * es;
            QTTS_HIP(hipMemcpy2DAsync(il.p, 2 * grp, m.mat(p + "mlp.gate_proj.weight"), grp, grp, (size_t)c.inter / 4, hipMemcpyDeviceToDevice, st));
            QTTS_HIP(hipMemcpy2DAsync((char*)il.p + grp, 2 * grp, m.mat(p + "mlp.up_proj.weight"), grp, grp, (size_t)c.inter / 4, hipMemcpyDeviceToDevice, st));
            w.gate_up_il = pack_matrix(m, il.p, 2 * (int64_t)c.inter, c.hidden);
            if (w.gate_up_il.packed) staging.push_back(std::move(il));
            else m.owned.push_back(std::move(il));
        }
        w.qkv = pack_matrix(m, qkv.p, nq + 2 * nk, c.hidden);
        w.gate_up = pack_matrix(m, gu.p, 2 * (int64_t)c.inter, c.hidden);
        if (w.qkv.packed) staging.push_back(std::move(qkv));
        else m.owned.push_back(std::move(qkv));
        if (w.gate_up.packed) staging.push_back(std::move(gu));
        else m.owned.push_back(std::move(gu));
        w.o = pack_raw(m, p + "self_attn.o_proj.weight", release);
        w.down = pack_raw(m, p + "mlp.down_proj.weight", release);
        if (qk_norm) {
            w.q_norm = m.f32(p + "self_attn.q_norm.weight");
            w.k_norm = m.f32(p + "self_attn.k_norm.weight");
        }
        if (layer_scale) {  // audio/tokenizer/v2/transformer.rs:279-300
            w.ls_attn = m.f32(p + "self_attn_layer_scale.scale");
            w.ls_mlp = m.f32(p + "mlp_layer_scale.scale");
        }
        w.ln1 = m.f32(p + "input_layernorm.weight");
        w.ln2 = m.f32(p + "post_attention_layernorm.weight");
        if (c.bias) {
            DevBuf qb;
            qb.alloc((size_t)(nq + 2 * nk) * 4);
            QTTS_HIP(hipMemcpyAsync(qb.p, m.f32(p + "self_attn.q_proj.bias"), (size_t)nq * 4, hipMemcpyDeviceToDevice, st));
            QTTS_HIP(hipMemcpyAsync((char*)qb.p + nq * 4, m.f32(p + "self_attn.k_proj.bias"), (size_t)nk * 4, hipMemcpyDeviceToDevice, st));
            QTTS_HIP(hipMemcpyAsync((char*)qb.p + (nq + nk) * 4, m.f32(p + "self_attn.v_proj.bias"), (size_t)nk * 4, hipMemcpyDeviceToDevice, st));
            w.qkv_bias = qb.as<float>();
            m.owned.push_back(std::move(qb));
            w.o_bias = m.f32(p + "self_attn.o_proj.bias");
        }
    }
    QTTS_HIP(hipStreamSynchronize(st));
    staging.clear();
    for (const std::string& k : release) m.raw.erase(k);
    // the separate q/k/v and gate/up copies are no longer needed
    for (int l = 0; l < c.layers; ++l) {
        const std::string p = prefix + "layers." + std::to_string(l) + ".";
        for (const char* n : {"self_attn.q_proj.weight", "self_attn.k_proj.weight", "self_attn.v_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight"})
            m.raw.erase(p + n);
    }
    s.final_norm = m.f32(prefix + "norm.weight");
}

static void stack_view(const StackW& b, StackW& v) {
    v.cfg = b.cfg;
    v.layers = b.layers;
    v.final_norm = b.final_norm;
    v.max_pos = 0;  // RoPE tables are grown on demand by the stream that needs them: each stream has its own
}

void Model::init_view(const Model& base) {
    QTTS_CHECK(base.finalized, ERR_STATE, "session: the model is not finalized");
    QTTS_CHECK(!finalized && raw.empty(), ERR_STATE, "session: the handle already holds a model");
    QTTS_HIP(hipSetDevice(ctx->device));
    is_view = true;
    locks = base.locks;
    cp_memo = base.cp_memo;
    tuning = base.tuning;
    tuning->latch(TUNE_SHARED);  // a session is opened: unless its first generate call came earlier, this weight set decodes with the kernels picked for several streams
    dtype = base.dtype;
    cfg = base.cfg;
    codec = base.codec;
    stack_view(base.talker, talker);
    stack_view(base.cp, cp);
    codec_emb = base.codec_emb;
    text_emb = base.text_emb;
    codec_head = base.codec_head;
    fc1 = base.fc1; fc2 = base.fc2; proj = base.proj;
    fc1_b = base.fc1_b; fc2_b = base.fc2_b; proj_b = base.proj_b;
    cp_emb = base.cp_emb;
    cp_head = base.cp_head;
    if (!cp_emb.empty()) {
        cp_emb_ptrs.alloc(cp_emb.size() * sizeof(void*));
        copy_sync(cp_emb_ptrs.p, cp_emb.data(), cp_emb.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream);
    }
    codec_state = base.codec_state;
    spk_state = base.spk_state;
    enc_state = base.enc_state;
    if (!talker.layers.empty()) {
        ensure_rope(talker, 1024);
        ensure_rope(cp, 64);
    }
    finalized = true;
}

void Model::finalize() {
    QTTS_CHECK(!finalized, ERR_STATE, "finalize called twice");
    QTTS_HIP(hipSetDevice(ctx->device));
    const bool have_talker = has("talker.model.norm.weight");
    if (have_talker) {
        build_stack(*this, talker, cfg.talker, "talker.model.", true, false, true);
        build_stack(*this, cp, cfg.cp, "talker.code_predictor.model.", true, false, true);
        const int H = cfg.talker.hidden, Ht = cfg.text_hidden, Hc = cfg.cp.hidden, G = cfg.groups;
        expect_shape(get("talker.model.codec_embedding.weight"), "talker.model.codec_embedding.weight", {cfg.vocab, H});
        expect_shape(get("talker.model.text_embedding.weight"), "talker.model.text_embedding.weight", {cfg.text_vocab, Ht});
        expect_shape(get("talker.codec_head.weight"), "talker.codec_head.weight", {cfg.vocab, H});
        expect_shape(get("talker.text_projection.linear_fc1.weight"), "talker.text_projection.linear_fc1.weight", {Ht, Ht});
        expect_shape(get("talker.text_projection.linear_fc2.weight"), "talker.text_projection.linear_fc2.weight", {H, Ht});
        codec_emb = mat("talker.model.codec_embedding.weight");
        text_emb = mat("talker.model.text_embedding.weight");
        std::vector<std::string> release;
        codec_head = pack_raw(*this, "talker.codec_head.weight", release);
        fc1 = pack_raw(*this, "talker.text_projection.linear_fc1.weight", release);
        fc2 = pack_raw(*this, "talker.text_projection.linear_fc2.weight", release);
        fc1_b = f32("talker.text_projection.linear_fc1.bias");
        fc2_b = f32("talker.text_projection.linear_fc2.bias");
        for (int j = 0; j < G - 1; ++j) {
            const std::string e = "talker.code_predictor.model.codec_embedding." + std::to_string(j) + ".weight";
            const std::string h = "talker.code_predictor.lm_head." + std::to_string(j) + ".weight";
            expect_shape(get(e), e, {cfg.cp_vocab, H});
            expect_shape(get(h), h, {cfg.cp_vocab, Hc});
            cp_emb.push_back(mat(e));
            cp_head.push_back(pack_raw(*this, h, release));
        }
        cp_emb_ptrs.alloc(cp_emb.size() * sizeof(void*));
        copy_sync(cp_emb_ptrs.p, cp_emb.data(), cp_emb.size() * sizeof(void*), hipMemcpyHostToDevice, ctx->stream);
        if (Hc != H) {  // code_predictor.rs:218-226
            expect_shape(get("talker.code_predictor.small_to_mtp_projection.weight"), "small_to_mtp_projection.weight", {Hc, H});
            proj = pack_raw(*this, "talker.code_predictor.small_to_mtp_projection.weight", release);
            proj_b = f32("talker.code_predictor.small_to_mtp_projection.bias");
        }
        QTTS_CHECK(cfg.vocab <= 4096 && cfg.cp_vocab <= 4096, ERR_SHAPE, "vocabulary exceeds the sampler's LDS staging (4096)");
        QTTS_CHECK(cfg.talker.hidden % 64 == 0 && cfg.cp.hidden % 64 == 0 && cfg.text_hidden % 64 == 0 &&
                       cfg.talker.inter % 64 == 0 && cfg.cp.inter % 64 == 0,
                   ERR_SHAPE, "hidden / intermediate sizes must be multiples of 64");
        QTTS_HIP(hipStreamSynchronize(ctx->stream));
        for (const std::string& k : release) raw.erase(k);
        ensure_rope(talker, 1024);
        ensure_rope(cp, 64);
    }
    finalize_codec(*this);
    finalize_speaker(*this);
    finalize_encoder(*this);
    finalized = true;
}

}  // namespace qtts
