// CPU oracle (test infrastructure): codec decoder, speaker encoder, mel front end.
#pragma once
#include "oracle.hpp"

namespace orc {

struct CodecConfig {  // audio/decoder/config.rs:13-97
    int codebook_size = 2048, hidden = 1024, latent = 1024, codebook_dim = 256;
    double rope_theta = 10000.0;
    int heads = 16, kv_heads = 16, head_dim = 0;
    bool attention_bias = false;
    int window = 72, inter = 3072;
    float eps = 1e-5f;
    int layers = 8, num_quantizers = 16;
    std::vector<int> upsample_rates{8, 5, 4, 3};
    std::vector<int> upsampling_ratios{2, 2};
    int decoder_dim = 1536;
};

struct CodecTrace {
    std::vector<float> after_quantizer, after_pre_conv, after_transformer, after_upsample;
    std::vector<std::vector<float>> after_block;
};

struct CodecModel {
    CodecConfig cfg;
    Weights w;
    int head_dim = 64;
    std::vector<std::vector<float>> codebooks;  // normalised, [Q][size][codebook_dim/2]
    void finalize();
    void transformer(std::vector<float>& x, int T) const;
    void convnext(std::vector<float>& x, int C, int T, const std::string& prefix) const;
    void decode(const uint32_t* codes, int T, std::vector<float>& wav, CodecTrace* trace) const;
    void chunked_decode(const uint32_t* codes, int T, int chunk, int left_ctx, std::vector<float>& wav) const;
    void decode_output(const uint32_t* codes, int len, std::vector<float>& wav) const;
};

struct SpeakerConfig {  // config/speaker_encoder_config.rs:12-76
    int mel_dim = 128, enc_dim = 1024;
    std::vector<int> channels{512, 512, 512, 512, 1536};
    std::vector<int> kernels{5, 3, 3, 3, 1};
    std::vector<int> dilations{1, 2, 3, 4, 1};
    int attention_channels = 128, res2net_scale = 8, se_channels = 128;
};

struct SpeakerModel {
    SpeakerConfig cfg;
    Weights w;  // keys start with "speaker_encoder."
    // SpeakerEncoder::forward (nn/speaker_encoder.rs:565-735): mel is [T][mel_dim], out [enc_dim]
    void forward(const float* mel, int T, float* out) const;
};

// TokenizerV2Encoder::encode (audio/encoder/v2.rs:87-120): 24 kHz waveform -> codes [T][valid_q]; weights under "encoder."
void encode_audio(const Weights& w, const float* audio, int n, int valid_q, std::vector<uint32_t>& codes, int& T_out,
                  std::vector<float>* margins = nullptr);

// audio/mel.rs:81-158 with the default MelSpectrogramConfig; returns [frames][128] (the layout
// Model::compute_mel_spectrogram hands to the speaker encoder).  n_frames is set.
void mel_spectrogram(const float* audio, int n, std::vector<float>& mel, int& n_frames);
void mel_filterbank(int n_fft, int num_mels, int sample_rate, double fmin, double fmax, std::vector<float>& fb);
void hann_window(int size, std::vector<float>& w);
void reflect_pad(const float* x, int n, int left, int right, std::vector<float>& out);

}  // namespace orc
