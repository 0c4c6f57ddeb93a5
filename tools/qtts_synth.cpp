// Host-side example in compiled code: everything through the C ABI of include/qtts.h, no Python.
//   qtts_synth <model_dir> <text_ids.txt> <out.wav> [--speaker NAME] [--language NAME] [--dtype f32|bf16] [--max-new-tokens N]
//              [--greedy] [--seed S] [--codes out_codes.txt]
// text_ids.txt holds the token ids of one utterance per line (whitespace-separated; the chat-template ids the reference's
// tokenize step produces, model/mod.rs tokenize_text).  The lines are decoded as ONE batch and written as <out>.wav (one line)
// or <out>.0.wav, <out>.1.wav, ...  The flow is that of qwen_tts_cli/src/main.rs:320-347 for custom-voice models:
// load -> generate -> decode_output -> 16-bit PCM WAV (`(sample * 32767.0) as i16` after a clamp to [-1, 1]).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "qtts.h"

static void die(const char* what) {
    std::fprintf(stderr, "qtts_synth: %s: %s\n", what, qtts_last_error());
    std::exit(1);
}

static void write_wav(const std::string& path, const float* x, int64_t n, int rate) {
    std::vector<int16_t> pcm((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i] < -1.f ? -1.f : (x[i] > 1.f ? 1.f : x[i]);
        pcm[(size_t)i] = (int16_t)(v * 32767.0f);  // truncation toward zero, like `as i16`
    }
    const uint32_t bytes = (uint32_t)(n * 2), riff = 36 + bytes, fmt = 16, brate = (uint32_t)rate * 2, r = (uint32_t)rate;
    const uint16_t pcm_tag = 1, ch = 1, align = 2, bits = 16;
    std::ofstream f(path, std::ios::binary);
    f.write("RIFF", 4); f.write((const char*)&riff, 4); f.write("WAVEfmt ", 8); f.write((const char*)&fmt, 4);
    f.write((const char*)&pcm_tag, 2); f.write((const char*)&ch, 2); f.write((const char*)&r, 4); f.write((const char*)&brate, 4);
    f.write((const char*)&align, 2); f.write((const char*)&bits, 2); f.write("data", 4); f.write((const char*)&bytes, 4);
    f.write((const char*)pcm.data(), bytes);
}

int main(int argc, char** argv) {
    if (argc < 4) {
        std::fprintf(stderr, "usage: qtts_synth <model_dir> <text_ids.txt> <out.wav> [--speaker S] [--language L] [--dtype f32|bf16] "
                             "[--max-new-tokens N] [--greedy] [--seed S] [--codes file]\n");
        return 2;
    }
    const std::string dir = argv[1], ids_path = argv[2], out = argv[3];
    std::string speaker, language = "auto", dtype = "bf16", codes_path;
    int max_new = 2048;
    bool greedy = false, seeded = false;
    uint64_t seed = 0;
    for (int i = 4; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* {
            if (i + 1 >= argc) { std::fprintf(stderr, "qtts_synth: %s needs a value\n", a.c_str()); std::exit(2); }
            return argv[++i];
        };
        if (a == "--speaker") speaker = next();
        else if (a == "--language") language = next();
        else if (a == "--dtype") dtype = next();
        else if (a == "--max-new-tokens") max_new = std::atoi(next());
        else if (a == "--greedy") greedy = true;
        else if (a == "--seed") { seed = std::strtoull(next(), nullptr, 10); seeded = true; }
        else if (a == "--codes") codes_path = next();
        else { std::fprintf(stderr, "qtts_synth: unknown option %s\n", a.c_str()); return 2; }
    }
    std::vector<std::vector<uint32_t>> texts;
    {
        std::ifstream f(ids_path);
        if (!f.good()) { std::fprintf(stderr, "qtts_synth: cannot read %s\n", ids_path.c_str()); return 1; }
        std::string line;
        while (std::getline(f, line)) {
            std::istringstream ss(line);
            std::vector<uint32_t> ids;
            uint64_t v;
            while (ss >> v) ids.push_back((uint32_t)v);
            if (!ids.empty()) texts.push_back(ids);
        }
    }
    if (texts.empty() || max_new < 1) { std::fprintf(stderr, "qtts_synth: no utterances\n"); return 1; }
    const int n = (int)texts.size();

    qtts_ctx* ctx = nullptr;
    if (qtts_init(0, &ctx) != 0) die("init");
    qtts_model* m = nullptr;
    if (qtts_model_load_dir(ctx, dir.c_str(), dtype == "f32" ? QTTS_F32 : QTTS_BF16, 1, 0, &m) != 0) die("load");
    int hidden = 0, vocab = 0, groups = 0, upsample = 0, enc_dim = 0;
    if (qtts_model_info(m, &hidden, &vocab, &groups, &upsample, &enc_dim) != 0) die("info");

    std::vector<qtts_request> reqs((size_t)n);
    for (int i = 0; i < n; ++i) {
        qtts_request& r = reqs[(size_t)i];
        std::memset(&r, 0, sizeof(r));
        r.text_ids = texts[(size_t)i].data();
        r.n_text = (int)texts[(size_t)i].size();
        r.language = language.c_str();
        r.speaker = speaker.empty() ? nullptr : speaker.c_str();
        r.non_streaming = 1;
        r.use_seed = seeded ? 1 : 0;
        r.seed = seed + (uint64_t)i;
    }
    // SamplingConfig::default (nn/sampling.rs:40-58), or `--greedy` as the reference CLI maps it (synthesis/synthesize_voice.rs:30-38)
    qtts_sampling s{};
    s.temperature = greedy ? 1.0 : 0.9; s.top_k = 50; s.top_p = 1.0; s.do_sample = greedy ? 0 : 1; s.repetition_penalty = 1.05;
    s.subtalker_do_sample = greedy ? 0 : 1; s.subtalker_temperature = 0.9; s.subtalker_top_k = 50; s.subtalker_top_p = 1.0;
    s.min_new_tokens = 2;
    std::vector<std::vector<uint32_t>> codes((size_t)n, std::vector<uint32_t>((size_t)max_new * groups));
    std::vector<qtts_codes_out> outs((size_t)n);
    for (int i = 0; i < n; ++i) outs[(size_t)i] = qtts_codes_out{codes[(size_t)i].data(), 0, 0};
    if (qtts_generate(m, reqs.data(), n, &s, max_new, outs.data()) != 0) die("generate");

    // decode_output: every row cut at its effective length, rows padded to the longest
    int T = 0;
    std::vector<int> lens((size_t)n);
    for (int i = 0; i < n; ++i) {
        lens[(size_t)i] = outs[(size_t)i].effective_len;
        T = lens[(size_t)i] > T ? lens[(size_t)i] : T;
    }
    if (!codes_path.empty()) {
        std::ofstream f(codes_path);
        for (int i = 0; i < n; ++i) {
            for (int t = 0; t < lens[(size_t)i]; ++t)
                for (int g = 0; g < groups; ++g) f << codes[(size_t)i][(size_t)t * groups + g] << (g + 1 < groups ? ' ' : '\n');
            f << '\n';
        }
    }
    if (T == 0) { std::fprintf(stderr, "qtts_synth: every utterance ended on its first frame\n"); return 1; }
    std::vector<uint32_t> batch((size_t)n * T * groups, 0u);
    for (int i = 0; i < n; ++i)
        std::memcpy(batch.data() + (size_t)i * T * groups, codes[(size_t)i].data(), (size_t)lens[(size_t)i] * groups * 4);
    const int64_t stride = (int64_t)T * upsample;
    std::vector<float> wav((size_t)n * stride);
    std::vector<int64_t> wav_lens((size_t)n);
    if (qtts_codec_decode(m, batch.data(), lens.data(), n, T, -2, 0, wav.data(), stride, wav_lens.data()) != 0) die("decode");
    for (int i = 0; i < n; ++i) {
        std::string path = out;
        if (n > 1) {
            const size_t dot = out.rfind('.');
            path = (dot == std::string::npos ? out : out.substr(0, dot)) + "." + std::to_string(i) + (dot == std::string::npos ? "" : out.substr(dot));
        }
        write_wav(path, wav.data() + (size_t)i * stride, wav_lens[(size_t)i], 24000);
        std::printf("%s: %d frames, %lld samples\n", path.c_str(), lens[(size_t)i], (long long)wav_lens[(size_t)i]);
    }
    qtts_model_free(m);
    qtts_shutdown(ctx);
    return 0;
}
