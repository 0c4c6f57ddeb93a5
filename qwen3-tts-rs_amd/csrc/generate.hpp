// Entry points implemented per subsystem (generate.cpp, codec.cpp, speaker.cpp).
#pragma once
#include "../../include/qtts.h"
#include "model.hpp"

namespace qtts {

void run_generate(Model& m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens, qtts_codes_out* outs,
                  const qtts_gen_debug* dbg);

// n utterances through `slots` rows: a finished row's slot is given to the next queued utterance (continuous batching)
void run_generate_continuous(Model& m, const qtts_request* reqs, int n, const qtts_sampling* s, int max_new_tokens, int slots,
                             qtts_codes_out* outs, const qtts_gen_debug* dbg);

void finalize_codec(Model& m);
void run_codec_decode(Model& m, const uint32_t* codes, const int* lens, int B, int T_max, int chunk, int left_ctx, float* wav_out,
                      int64_t wav_stride, int64_t* wav_lens);

struct CodecStream;  // streaming decode session (codec.cpp)
CodecStream* codec_stream_open(Model& m, int B, int max_chunk);
void codec_stream_push(CodecStream& sc, const uint32_t* codes, int n, float* wav_out, int64_t wav_stride, int64_t* n_samples);
void codec_stream_close(CodecStream* sc);

void finalize_encoder(Model& m);
void run_encode_audio(Model& m, const float* audio, const int* n_samples, int B, int max_samples, int Q, uint32_t* codes_out, int T_cap,
                      int* T_out);

void finalize_speaker(Model& m);
void run_mel(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride,
             int* n_frames);
// the same without any allocation: staging in a fixed per-GPU device array (speaker.cpp); mel_scratch looks the array's
// device address up once per GPU and must be called with that GPU current
void run_mel_noalloc(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* mel_out, int frames_stride,
                     int* n_frames);
unsigned char* mel_scratch(int device);
size_t mel_scratch_bytes();
void run_speaker_embed(Model& m, const float* audio, const int* n_samples, int B, int max_samples, float* emb_out);
void run_speaker_embed_mel(Model& m, const float* mel, const int* n_frames, int B, int frames_stride, float* emb_out);

}  // namespace qtts
