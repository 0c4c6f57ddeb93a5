// CPU oracle (test infrastructure): flat C entry points for the Python tests / bench baseline.
// Configs arrive as arrays of doubles whose order is fixed by tests/oracle_bind.py.
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "oracle_codec.hpp"

using namespace orc;

namespace {
struct Handle {
    Model model;
    CodecModel codec;
    SpeakerModel spk;
    Weights enc;  // codec encoder tensors ("encoder.*")
    std::string err;
};
thread_local std::string g_err;

AttnShape attn_from(const double* c) {
    AttnShape a;
    a.hidden = (int)c[0]; a.inter = (int)c[1]; a.layers = (int)c[2]; a.nh = (int)c[3]; a.nkv = (int)c[4]; a.d = (int)c[5];
    a.eps = (float)c[6]; a.theta = c[7]; a.window = (int)c[8]; a.bias = c[9] != 0.0;
    return a;
}
}  // namespace

#define ORC_TRY try {
#define ORC_CATCH                                   \
    }                                               \
    catch (const std::exception& e) {               \
        g_err = e.what();                           \
        return -1;                                  \
    }                                               \
    return 0;

extern "C" {

const char* orc_last_error() { return g_err.c_str(); }

void* orc_create() { return new Handle(); }
void orc_free(void* h) { delete (Handle*)h; }

int orc_load_tensor(void* hv, const char* key, const float* data, const int64_t* shape, int rank) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    Tensor t;
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    t.v.assign(data, data + n);
    std::string k(key);
    if (k.rfind("decoder.", 0) == 0) h->codec.w[k] = std::move(t);
    else if (k.rfind("encoder.", 0) == 0) h->enc[k] = std::move(t);
    else if (k.rfind("speaker_encoder.", 0) == 0) h->spk.w[k] = std::move(t);
    else h->model.w[k] = std::move(t);
    ORC_CATCH
}

// talker[10], cp[10], misc[15]: vocab,text_hidden,text_vocab,groups,cp_vocab,eos,bos,pad,think,nothink,think_bos,
// think_eos,tts_bos,tts_eos,tts_pad
int orc_set_model_config(void* hv, const double* talker, const double* cp, const double* misc) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    ModelConfig& c = h->model.cfg;
    c.talker = attn_from(talker);
    c.cp = attn_from(cp);
    c.vocab = (int)misc[0]; c.text_hidden = (int)misc[1]; c.text_vocab = (int)misc[2]; c.groups = (int)misc[3];
    c.cp_vocab = (int)misc[4]; c.eos = (int)misc[5]; c.bos = (int)misc[6]; c.pad = (int)misc[7]; c.think = (int)misc[8];
    c.nothink = (int)misc[9]; c.think_bos = (int)misc[10]; c.think_eos = (int)misc[11]; c.tts_bos = (int)misc[12];
    c.tts_eos = (int)misc[13]; c.tts_pad = (int)misc[14];
    h->model.finalize();
    ORC_CATCH
}

// cfg[14]: codebook_size,hidden,latent,codebook_dim,rope_theta,heads,kv_heads,head_dim,attention_bias,window,inter,eps,
// layers,num_quantizers ; then decoder_dim ; rates / ratios arrays
int orc_set_codec_config(void* hv, const double* c, int decoder_dim, const int* rates, int n_rates, const int* ratios,
                         int n_ratios) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    CodecConfig& g = h->codec.cfg;
    g.codebook_size = (int)c[0]; g.hidden = (int)c[1]; g.latent = (int)c[2]; g.codebook_dim = (int)c[3]; g.rope_theta = c[4];
    g.heads = (int)c[5]; g.kv_heads = (int)c[6]; g.head_dim = (int)c[7]; g.attention_bias = c[8] != 0.0; g.window = (int)c[9];
    g.inter = (int)c[10]; g.eps = (float)c[11]; g.layers = (int)c[12]; g.num_quantizers = (int)c[13];
    g.decoder_dim = decoder_dim;
    g.upsample_rates.assign(rates, rates + n_rates);
    g.upsampling_ratios.assign(ratios, ratios + n_ratios);
    h->codec.finalize();
    ORC_CATCH
}

int orc_set_speaker_config(void* hv, int mel_dim, int enc_dim, const int* channels, const int* kernels, const int* dilations,
                           int n, int attention_channels, int res2net_scale, int se_channels) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    SpeakerConfig& s = h->spk.cfg;
    s.mel_dim = mel_dim; s.enc_dim = enc_dim;
    s.channels.assign(channels, channels + n);
    s.kernels.assign(kernels, kernels + n);
    s.dilations.assign(dilations, dilations + n);
    s.attention_channels = attention_channels; s.res2net_scale = res2net_scale; s.se_channels = se_channels;
    ORC_CATCH
}

static SamplingConfig sampling_from(const double* s) {
    SamplingConfig c;
    c.temperature = s[0]; c.top_k = (int)s[1]; c.top_p = s[2]; c.do_sample = (int)s[3]; c.repetition_penalty = s[4];
    c.min_new_tokens = (int)s[5]; c.eos_token_id = (int)s[6]; c.suppress_lo = (int)s[7]; c.suppress_hi = (int)s[8];
    c.subtalker_do_sample = (int)s[9]; c.subtalker_temperature = s[10]; c.subtalker_top_k = (int)s[11];
    c.subtalker_top_p = s[12];
    return c;
}

// One utterance through ConditionalGeneration::generate.  ints[]: n_text, n_instruct, n_prefill, speaker_token,
// has_spk_embed, non_streaming, n_ref_text, ref_T, max_new_tokens, use_seed, trace, rng_stride, rng_offset (the utterance is
// row rng_offset of rng_stride rows that share the stream of `seed`: MT19937::stride; 1, 0 = a stream of its own).
int orc_generate(void* hv, const uint32_t* text_ids, const uint32_t* instruct_ids, const int* prefill,
                 const float* spk_embed, const uint32_t* ref_text_ids, const uint32_t* ref_codes, const int* ints,
                 const double* sampling, uint64_t seed, uint32_t* codes_out, int* num_steps, int* eff_len,
                 float* logits_trace, float* hidden_trace, float* min_margin) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    GenRequest r;
    r.text_ids.assign(text_ids, text_ids + ints[0]);
    if (ints[1] > 0) r.instruct_ids.assign(instruct_ids, instruct_ids + ints[1]);
    r.codec_prefill.assign(prefill, prefill + ints[2]);
    r.speaker_token = ints[3];
    if (ints[4]) r.speaker_embed.assign(spk_embed, spk_embed + h->model.cfg.talker.hidden);
    r.non_streaming = ints[5] != 0;
    if (ints[6] > 0) r.ref_text_ids.assign(ref_text_ids, ref_text_ids + ints[6]);
    r.ref_T = ints[7];
    if (ints[7] > 0) r.ref_codes.assign(ref_codes, ref_codes + (size_t)ints[7] * h->model.cfg.groups);
    SamplingConfig s = sampling_from(sampling);
    MT19937 rng((uint32_t)seed);
    rng.stride = ints[11] > 0 ? ints[11] : 1;
    rng.offset = ints[12];
    GenOutput out;
    h->model.generate(r, s, ints[8], &rng, ints[10] != 0, out);
    std::memcpy(codes_out, out.codes.data(), out.codes.size() * sizeof(uint32_t));
    *num_steps = out.num_steps;
    *eff_len = out.effective_len;
    if (ints[10] && logits_trace) std::memcpy(logits_trace, out.logits_trace.data(), out.logits_trace.size() * sizeof(float));
    if (ints[10] && hidden_trace) std::memcpy(hidden_trace, out.hidden_trace.data(), out.hidden_trace.size() * sizeof(float));
    if (min_margin) *min_margin = out.min_margin;
    ORC_CATCH
}

// prompt embeddings only (build_prompt): returns P and writes [P][hidden] (cap rows), trailing length in *Tt
int orc_build_prompt(void* hv, const uint32_t* text_ids, const uint32_t* instruct_ids, const int* prefill,
                     const float* spk_embed, const uint32_t* ref_text_ids, const uint32_t* ref_codes, const int* ints,
                     float* prompt_out, int cap_rows, int* P, float* trailing_out, int cap_trailing, int* Tt) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    GenRequest r;
    r.text_ids.assign(text_ids, text_ids + ints[0]);
    if (ints[1] > 0) r.instruct_ids.assign(instruct_ids, instruct_ids + ints[1]);
    r.codec_prefill.assign(prefill, prefill + ints[2]);
    r.speaker_token = ints[3];
    if (ints[4]) r.speaker_embed.assign(spk_embed, spk_embed + h->model.cfg.talker.hidden);
    r.non_streaming = ints[5] != 0;
    if (ints[6] > 0) r.ref_text_ids.assign(ref_text_ids, ref_text_ids + ints[6]);
    r.ref_T = ints[7];
    if (ints[7] > 0) r.ref_codes.assign(ref_codes, ref_codes + (size_t)ints[7] * h->model.cfg.groups);
    std::vector<float> prompt, trailing, pad;
    int p = 0, tt = 0;
    h->model.build_prompt(r, prompt, p, trailing, tt, pad);
    if (p > cap_rows || tt > cap_trailing) throw std::runtime_error("oracle: prompt buffer too small");
    std::memcpy(prompt_out, prompt.data(), prompt.size() * sizeof(float));
    if (trailing_out) std::memcpy(trailing_out, trailing.data(), trailing.size() * sizeof(float));
    *P = p;
    *Tt = tt;
    ORC_CATCH
}

// Talker over T given embeddings starting from an empty cache: returns post-norm hidden [T][hidden] and logits [T][vocab].
int orc_talker_forward(void* hv, const float* embeds, int T, float* hidden_out, float* logits_out) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    const ModelConfig& c = h->model.cfg;
    std::vector<KVCacheLayer> cache((size_t)c.talker.layers);
    for (auto& k : cache) {
        k.cap = T;
        k.k.assign((size_t)c.talker.nkv * T * c.talker.d, 0.f);
        k.v.assign((size_t)c.talker.nkv * T * c.talker.d, 0.f);
    }
    std::vector<float> x(embeds, embeds + (size_t)T * c.talker.hidden), hidden;
    h->model.talker_forward(x.data(), T, 0, cache, hidden);
    std::memcpy(hidden_out, hidden.data(), hidden.size() * sizeof(float));
    if (logits_out)
        for (int t = 0; t < T; ++t) h->model.talker_logits(&hidden[(size_t)t * c.talker.hidden], logits_out + (size_t)t * c.vocab);
    ORC_CATCH
}

// Code predictor for one frame: greedy or sampled (seeded) codes, plus all G-1 logit rows.
int orc_code_predictor(void* hv, const float* talker_hidden, int cb0, const double* sampling, uint64_t seed,
                       uint32_t* codes_out, float* logits_out) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    SamplingConfig s = sampling_from(sampling).for_subtalker();
    MT19937 rng((uint32_t)seed);
    std::vector<float> dbg;
    h->model.code_predictor(talker_hidden, cb0, s, &rng, codes_out, nullptr, &dbg);
    if (logits_out) std::memcpy(logits_out, dbg.data(), dbg.size() * sizeof(float));
    ORC_CATCH
}

// mode 0: TokenizerV2::decode; 1: chunked_decode(chunk, left_ctx); 2: Model::decode_output policy
int orc_codec_decode(void* hv, const uint32_t* codes, int T, int mode, int chunk, int left_ctx, float* wav, int64_t cap,
                     int64_t* n_out) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    std::vector<float> w;
    if (mode == 0) h->codec.decode(codes, T, w, nullptr);
    else if (mode == 1) h->codec.chunked_decode(codes, T, chunk, left_ctx, w);
    else h->codec.decode_output(codes, T, w);
    if ((int64_t)w.size() > cap) throw std::runtime_error("oracle: wav buffer too small");
    std::memcpy(wav, w.data(), w.size() * sizeof(float));
    *n_out = (int64_t)w.size();
    ORC_CATCH
}

// Intermediate activations of one full decode, channels-first, for stage-by-stage parity.
// stage: 0 after_quantizer [Dc][T], 1 after_pre_conv, 2 after_transformer, 3 after_upsample, 4+b after vocoder block b
int orc_codec_trace(void* hv, const uint32_t* codes, int T, int stage, float* out, int64_t cap, int64_t* n_out) {
    ORC_TRY
    Handle* h = (Handle*)hv;
    std::vector<float> w;
    CodecTrace tr;
    h->codec.decode(codes, T, w, &tr);
    const std::vector<float>* src = nullptr;
    if (stage == 0) src = &tr.after_quantizer;
    else if (stage == 1) src = &tr.after_pre_conv;
    else if (stage == 2) src = &tr.after_transformer;
    else if (stage == 3) src = &tr.after_upsample;
    else if (stage - 4 < (int)tr.after_block.size()) src = &tr.after_block[(size_t)stage - 4];
    if (!src) throw std::runtime_error("oracle: bad trace stage");
    if ((int64_t)src->size() > cap) throw std::runtime_error("oracle: trace buffer too small");
    std::memcpy(out, src->data(), src->size() * sizeof(float));
    *n_out = (int64_t)src->size();
    ORC_CATCH
}

int orc_encode_audio(void* hv, const float* audio, int n, int valid_q, uint32_t* codes, int64_t cap, int* T_out) {
    ORC_TRY
    std::vector<uint32_t> c;
    int T = 0;
    encode_audio(((Handle*)hv)->enc, audio, n, valid_q, c, T);
    if ((int64_t)c.size() > cap) throw std::runtime_error("oracle: code buffer too small");
    std::memcpy(codes, c.data(), c.size() * sizeof(uint32_t));
    *T_out = T;
    ORC_CATCH
}

// same, plus margins [T][valid_q][2]: (second nearest - nearest distance, magnitude of the nearest distance's terms)
int orc_encode_audio_margins(void* hv, const float* audio, int n, int valid_q, uint32_t* codes, float* margins, int64_t cap, int* T_out) {
    ORC_TRY
    std::vector<uint32_t> c;
    std::vector<float> mg;
    int T = 0;
    encode_audio(((Handle*)hv)->enc, audio, n, valid_q, c, T, &mg);
    if ((int64_t)c.size() > cap) throw std::runtime_error("oracle: code buffer too small");
    std::memcpy(codes, c.data(), c.size() * sizeof(uint32_t));
    std::memcpy(margins, mg.data(), mg.size() * sizeof(float));
    *T_out = T;
    ORC_CATCH
}

int orc_speaker_embed(void* hv, const float* mel, int T, float* out) {
    ORC_TRY
    ((Handle*)hv)->spk.forward(mel, T, out);
    ORC_CATCH
}

int orc_mel(const float* audio, int n, float* out, int64_t cap, int* n_frames) {
    ORC_TRY
    std::vector<float> m;
    int f = 0;
    mel_spectrogram(audio, n, m, f);
    if ((int64_t)m.size() > cap) throw std::runtime_error("oracle: mel buffer too small");
    std::memcpy(out, m.data(), m.size() * sizeof(float));
    *n_frames = f;
    ORC_CATCH
}

// ---- micro known-answer helpers -------------------------------------------------------
void orc_mt_u32(uint32_t seed, int n, uint32_t* out) {
    MT19937 r(seed);
    for (int i = 0; i < n; ++i) out[i] = r.next_u32();
}
void orc_mt_uniform(uint32_t seed, int n, float* out) {
    MT19937 r(seed);
    for (int i = 0; i < n; ++i) out[i] = r.uniform();
}
// row `offset` of `stride` rows that share the stream of `seed` (MT19937::stride)
void orc_mt_uniform_strided(uint32_t seed, int n, int stride, int offset, float* out) {
    MT19937 r(seed);
    r.stride = stride;
    r.offset = offset;
    for (int i = 0; i < n; ++i) out[i] = r.uniform();
}
void orc_randn_vectorized(uint32_t seed, int n, float* out) {
    MT19937 r(seed);
    randn_vectorized(r, out, n);
}
// processors: which = 0 rep-penalty (arg = penalty, hist), 1 temperature, 2 top-k, 3 top-p,
// 4 suppress (hist = ids to suppress; apply_suppress_tokens, sampling.rs:143-170)
void orc_processor(int which, float* logits, int n, double arg, const int64_t* hist, int n_hist) {
    std::vector<float> l(logits, logits + n);
    std::vector<int64_t> hv(hist, hist + n_hist);
    if (which == 0) apply_repetition_penalty(l, hv, arg);
    else if (which == 1) apply_temperature(l, arg);
    else if (which == 2) apply_top_k(l, (int)arg);
    else if (which == 4) {
        for (int64_t id : hv)
            if (id >= 0 && id < n) l[(size_t)id] = -INFINITY;
    } else apply_top_p(l, arg);
    std::memcpy(logits, l.data(), sizeof(float) * (size_t)n);
}
int64_t orc_sample_token(const float* logits, int n, const double* sampling, const int64_t* hist, int n_hist, uint32_t seed) {
    SamplingConfig s = sampling_from(sampling);
    std::vector<int64_t> hv(hist, hist + n_hist);
    MT19937 r(seed);
    return sample_token(logits, n, s, hv, &r);
}
void orc_hann(int n, float* out) {
    std::vector<float> w;
    hann_window(n, w);
    std::memcpy(out, w.data(), sizeof(float) * (size_t)n);
}
int orc_reflect_pad(const float* x, int n, int l, int r, float* out) {
    std::vector<float> o;
    reflect_pad(x, n, l, r, o);
    std::memcpy(out, o.data(), sizeof(float) * o.size());
    return (int)o.size();
}
void orc_mel_filterbank(int n_fft, int num_mels, int sr, double fmin, double fmax, float* out) {
    std::vector<float> fb;
    mel_filterbank(n_fft, num_mels, sr, fmin, fmax, fb);
    std::memcpy(out, fb.data(), sizeof(float) * fb.size());
}
// rotate_half (nn/rope/mod.rs:64-76): [-x2, x1] over the two halves of the last dimension
void orc_rotate_half(const float* x, int n, float* out) {
    const int half = n / 2;
    for (int i = 0; i < half; ++i) {
        out[i] = -x[i + half];
        out[i + half] = x[i];
    }
}
// create_position_ids_with_padding (nn/generation_utils.rs:132-153): max(0, i - pad_len) per row
void orc_position_ids_with_padding(const int* lengths, int n, int max_len, int64_t* out) {
    for (int b = 0; b < n; ++b) {
        const int pad = max_len - lengths[b];
        for (int i = 0; i < max_len; ++i) out[(size_t)b * max_len + i] = std::max(0, i - pad);
    }
}
// create_attention_mask_from_lengths (nn/generation_utils.rs:63-121): 0 where causal and both positions are
// non-padding, -inf elsewhere; rows of padding queries are fully masked
void orc_attention_mask_from_lengths(const int* lengths, int n, int max_len, float* out) {
    for (int b = 0; b < n; ++b) {
        const int pad = max_len - lengths[b];
        for (int i = 0; i < max_len; ++i)
            for (int j = 0; j < max_len; ++j) {
                const bool ok = i >= j && i >= pad && j >= pad;
                out[((size_t)b * max_len + i) * max_len + j] = ok ? 0.0f : -INFINITY;
            }
    }
}
void orc_set_round_probe(int mask) { set_round_probe(mask); }
void orc_linear(const float* x, const float* w, const float* b, float* y, int64_t M, int64_t N, int64_t K) { linear(x, w, b, y, M, N, K); }
void orc_rmsnorm(const float* x, const float* w, float* y, int64_t rows, int64_t n, float eps) { rmsnorm(x, w, y, rows, n, eps); }

}  // extern "C"
