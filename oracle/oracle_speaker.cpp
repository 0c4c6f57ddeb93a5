// CPU oracle (test infrastructure): ECAPA-TDNN speaker encoder and log-mel front end.
// Restates qwen_tts/src/nn/speaker_encoder.rs and qwen_tts/src/audio/mel.rs.  Pinned on transformers' ECAPA_TimeDelayNet
// (tests/golden/make_ecapa_golden.py, 2e-6 max-abs) and on a torch.stft pipeline (tests/test_oracle_golden.py).
#include "oracle_codec.hpp"

#include <algorithm>
#include <complex>
#include <stdexcept>

namespace orc {

using Mat = std::vector<float>;  // [C][T]

// TimeDelayNetBlock (speaker_encoder.rs:20-108): reflect "same" padding, conv1d(dilation), ReLU.
static void tdnn(const Weights& w, const std::string& p, const Mat& x, int Cin, int T, int k, int dil, Mat& y, int& Cout,
                 bool relu = true) {
    const Tensor& wt = W(w, p + "conv.weight");
    const float* b = W(w, p + "conv.bias").p();
    Cout = (int)wt.dim(0);
    if ((int)wt.dim(1) != Cin || (int)wt.dim(2) != k) throw std::runtime_error("oracle: tdnn shape mismatch " + p);
    const int pad = (k - 1) * dil / 2;
    if (pad >= T && pad > 0) throw std::runtime_error("oracle: reflect pad longer than the signal");
    y.assign((size_t)Cout * T, 0.f);
#pragma omp parallel
    {
        std::vector<float> patch((size_t)Cin * k);
#pragma omp for schedule(static)
        for (int t = 0; t < T; ++t) {
            for (int ci = 0; ci < Cin; ++ci)
                for (int kk = 0; kk < k; ++kk) {
                    int xi = t + kk * dil - pad;  // index into the unpadded signal
                    if (xi < 0) xi = -xi;                      // reflect_pad_1d :81-83
                    else if (xi >= T) xi = 2 * (T - 1) - xi;   // :91-93
                    patch[(size_t)ci * k + kk] = x[(size_t)ci * T + xi];
                }
            for (int co = 0; co < Cout; ++co) {
                float s = dot(patch.data(), wt.p() + (size_t)co * Cin * k, (int64_t)Cin * k) + b[co];
                y[(size_t)co * T + t] = relu ? std::max(s, 0.f) : s;
            }
        }
    }
}

void SpeakerModel::forward(const float* mel, int T, float* out) const {
    const auto& c = cfg;
    const std::string P = "speaker_encoder.";
    Mat x((size_t)c.mel_dim * T);
    for (int t = 0; t < T; ++t)
        for (int m = 0; m < c.mel_dim; ++m) x[(size_t)m * T + t] = mel[(size_t)t * c.mel_dim + m];
    Mat h;
    int C;
    tdnn(w, P + "blocks.0.", x, c.mel_dim, T, c.kernels[0], c.dilations[0], h, C);
    std::vector<Mat> outs;
    const int nblk = (int)c.channels.size() - 2;
    for (int i = 1; i <= nblk; ++i) {  // SqueezeExcitationRes2NetBlock :299-315
        const std::string B = P + "blocks." + std::to_string(i) + ".";
        Mat a, r, t2;
        int C1;
        tdnn(w, B + "tdnn1.", h, C, T, 1, 1, a, C1);
        // Res2NetBlock :156-183
        const int sc = c.res2net_scale, cs = C1 / sc;
        r.assign((size_t)C1 * T, 0.f);
        Mat prev;
        for (int j = 0; j < sc; ++j) {
            Mat part(a.begin() + (size_t)j * cs * T, a.begin() + (size_t)(j + 1) * cs * T), o;
            if (j == 0) o = part;
            else {
                if (j >= 2)
                    for (size_t e = 0; e < part.size(); ++e) part[e] = part[e] + prev[e];
                int co;
                tdnn(w, B + "res2net_block.blocks." + std::to_string(j - 1) + ".", part, cs, T, c.kernels[(size_t)i],
                     c.dilations[(size_t)i], o, co);
            }
            std::copy(o.begin(), o.end(), r.begin() + (size_t)j * cs * T);
            prev.swap(o);
        }
        int C2;
        tdnn(w, B + "tdnn2.", r, C1, T, 1, 1, t2, C2);
        // SqueezeExcitationBlock :218-230
        std::vector<float> mean((size_t)C2), s1, s2;
        for (int ch = 0; ch < C2; ++ch) {
            float s = 0.f;
            for (int t = 0; t < T; ++t) s += t2[(size_t)ch * T + t];
            mean[(size_t)ch] = s / (float)T;
        }
        const Tensor& w1 = W(w, B + "se_block.conv1.weight");
        const Tensor& w2 = W(w, B + "se_block.conv2.weight");
        s1.resize((size_t)w1.dim(0));
        s2.resize((size_t)w2.dim(0));
        linear(mean.data(), w1.p(), W(w, B + "se_block.conv1.bias").p(), s1.data(), 1, w1.dim(0), C2);
        for (auto& v : s1) v = std::max(v, 0.f);
        linear(s1.data(), w2.p(), W(w, B + "se_block.conv2.bias").p(), s2.data(), 1, w2.dim(0), w1.dim(0));
        for (auto& v : s2) v = 1.0f / (1.0f + std::exp(-v));
        for (int ch = 0; ch < C2; ++ch)
            for (int t = 0; t < T; ++t) t2[(size_t)ch * T + t] = t2[(size_t)ch * T + t] * s2[(size_t)ch];
        if (C == C2)
            for (size_t e = 0; e < t2.size(); ++e) t2[e] = h[e] + t2[e];
        h = t2;
        C = C2;
        outs.push_back(h);
    }
    // MFA: cat(block outputs) -> TDNN
    Mat cat;
    int Ccat = 0;
    for (auto& o : outs) {
        cat.insert(cat.end(), o.begin(), o.end());
        Ccat += (int)(o.size() / (size_t)T);
    }
    Mat m;
    int Cm;
    tdnn(w, P + "mfa.", cat, Ccat, T, c.kernels.back(), c.dilations.back(), m, Cm);
    // AttentiveStatisticsPooling :398-456 with lengths = None
    auto stats = [&](const Mat& wts /* [Cw][T], Cw in {1, Cm} */, int Cw, std::vector<float>& mean, std::vector<float>& sd) {
        mean.assign((size_t)Cm, 0.f);
        sd.assign((size_t)Cm, 0.f);
        for (int ch = 0; ch < Cm; ++ch) {
            const float* wr = &wts[(size_t)(Cw == 1 ? 0 : ch) * T];
            float mu = 0.f;
            for (int t = 0; t < T; ++t) mu += m[(size_t)ch * T + t] * wr[t];
            float var = 0.f;
            for (int t = 0; t < T; ++t) {
                float d = m[(size_t)ch * T + t] - mu;
                var += d * d * wr[t];
            }
            mean[(size_t)ch] = mu;
            sd[(size_t)ch] = std::sqrt(var + 1e-12f);
        }
    };
    Mat uni((size_t)T, 1.0f / (float)T);
    std::vector<float> mu, sd;
    stats(uni, 1, mu, sd);
    Mat ain((size_t)3 * Cm * T);
    std::copy(m.begin(), m.end(), ain.begin());
    for (int ch = 0; ch < Cm; ++ch)
        for (int t = 0; t < T; ++t) {
            ain[(size_t)(Cm + ch) * T + t] = mu[(size_t)ch];
            ain[(size_t)(2 * Cm + ch) * T + t] = sd[(size_t)ch];
        }
    Mat att, att2;
    int Ca;
    tdnn(w, P + "asp.tdnn.", ain, 3 * Cm, T, 1, 1, att, Ca);
    for (auto& v : att) v = std::tanh(v);
    // asp.conv is a plain 1x1 conv (no ReLU): reuse tdnn() without the activation
    {
        const Tensor& wt = W(w, P + "asp.conv.weight");
        const float* b = W(w, P + "asp.conv.bias").p();
        att2.assign((size_t)Cm * T, 0.f);
        std::vector<float> col((size_t)Ca);
        for (int t = 0; t < T; ++t) {
            for (int ci = 0; ci < Ca; ++ci) col[(size_t)ci] = att[(size_t)ci * T + t];
            for (int co = 0; co < Cm; ++co) att2[(size_t)co * T + t] = dot(col.data(), wt.p() + (size_t)co * Ca, Ca) + b[co];
        }
    }
    for (int ch = 0; ch < Cm; ++ch) softmax_inplace(&att2[(size_t)ch * T], T);
    stats(att2, Cm, mu, sd);
    std::vector<float> pooled(mu);
    pooled.insert(pooled.end(), sd.begin(), sd.end());
    const Tensor& fw = W(w, P + "fc.weight");
    linear(pooled.data(), fw.p(), W(w, P + "fc.bias").p(), out, 1, fw.dim(0), 2 * Cm);
}

// ------------------------------------------------------------------------------------ mel
void hann_window(int size, std::vector<float>& w) {  // mel.rs:54-61 (f32 arithmetic)
    w.resize((size_t)size);
    for (int n = 0; n < size; ++n) {
        float x = 2.0f * 3.14159265358979323846f * (float)n / (float)size;
        w[(size_t)n] = 0.5f * (1.0f - std::cos(x));
    }
}

void reflect_pad(const float* x, int n, int left, int right, std::vector<float>& out) {  // mel.rs:161-181
    out.clear();
    for (int i = left; i >= 1; --i) out.push_back(x[i < n ? i : n - 1]);
    out.insert(out.end(), x, x + n);
    for (int i = 0; i < right; ++i) out.push_back(x[n >= 2 + i ? n - 2 - i : 0]);
}

void mel_filterbank(int n_fft, int num_mels, int sample_rate, double fmin, double fmax, std::vector<float>& fb) {
    // mel.rs:236-327 (Slaney scale, f64 construction, f32 storage, Slaney area normalisation)
    const int n_freqs = n_fft / 2 + 1;
    const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = std::log(6.4) / 27.0;
    auto hz_to_mel = [&](double hz) { return hz < min_log_hz ? hz / f_sp : min_log_mel + std::log(hz / min_log_hz) / logstep; };
    auto mel_to_hz = [&](double mel) { return mel < min_log_mel ? mel * f_sp : min_log_hz * std::exp((mel - min_log_mel) * logstep); };
    const double mmin = hz_to_mel(fmin), mmax = hz_to_mel(fmax);
    std::vector<double> hz((size_t)num_mels + 2);
    for (int i = 0; i <= num_mels + 1; ++i) hz[(size_t)i] = mel_to_hz(mmin + (mmax - mmin) * (double)i / (double)(num_mels + 1));
    fb.assign((size_t)num_mels * n_freqs, 0.f);
    for (int m = 0; m < num_mels; ++m) {
        const double fl = hz[(size_t)m], fc = hz[(size_t)m + 1], fr = hz[(size_t)m + 2];
        const double ld = fc - fl, ud = fr - fc;
        const float enorm = 2.0f / (float)(fr - fl);
        for (int b = 0; b < n_freqs; ++b) {
            const double freq = (double)b * (double)sample_rate / (double)n_fft;
            const double lower = ld > 0.0 ? (freq - fl) / ld : 0.0;
            const double upper = ud > 0.0 ? (fr - freq) / ud : 0.0;
            const double wgt = std::max(std::min(lower, upper), 0.0);
            fb[(size_t)m * n_freqs + b] = (float)wgt * enorm;
        }
    }
}

static void fft_pow2(std::vector<std::complex<double>>& a) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; ++i) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        const double ang = -2.0 * 3.14159265358979323846 / (double)len;
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                std::complex<double> wv(std::cos(ang * (double)k), std::sin(ang * (double)k));
                std::complex<double> u = a[i + k], v = a[i + k + len / 2] * wv;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
}

void mel_spectrogram(const float* audio, int n, std::vector<float>& mel, int& n_frames) {
    const int n_fft = 1024, hop = 256, num_mels = 128, sr = 24000, n_freqs = n_fft / 2 + 1;
    std::vector<float> fb, win, padded;
    mel_filterbank(n_fft, num_mels, sr, 0.0, sr / 2.0, fb);
    hann_window(n_fft, win);
    const int pad = (n_fft - hop) / 2;
    reflect_pad(audio, n, pad, pad, padded);
    if ((int)padded.size() < n_fft) throw std::runtime_error("oracle: audio shorter than one STFT frame");
    n_frames = ((int)padded.size() - n_fft) / hop + 1;
    mel.assign((size_t)n_frames * num_mels, 0.f);
#pragma omp parallel
    {
        std::vector<std::complex<double>> buf((size_t)n_fft);
        std::vector<float> mag((size_t)n_freqs);
#pragma omp for schedule(static)
        for (int f = 0; f < n_frames; ++f) {
            for (int i = 0; i < n_fft; ++i) buf[(size_t)i] = (double)(padded[(size_t)f * hop + i] * win[(size_t)i]);
            fft_pow2(buf);
            for (int b = 0; b < n_freqs; ++b) {
                float re = (float)buf[(size_t)b].real(), im = (float)buf[(size_t)b].imag();
                mag[(size_t)b] = std::sqrt(re * re + im * im + 1e-9f);
            }
            for (int m = 0; m < num_mels; ++m) {
                float s = dot(&fb[(size_t)m * n_freqs], mag.data(), n_freqs);
                mel[(size_t)f * num_mels + m] = std::log(std::max(s, 1e-5f));  // dynamic_range_compression :45-48
            }
        }
    }
}

}  // namespace orc
