// Load-time weight preparation shared by the codec decoder and the speaker encoder: download a tensor as f32,
// rearrange conv / transposed-conv / linear weights into GEMM rows [N][taps * CinP] with channel counts padded to
// multiples of 64, upload in the model dtype.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "model.hpp"

namespace qtts {

inline int pad64(int c) { return (c + 63) / 64 * 64; }

inline std::vector<float> host_f32(const Model& m, const std::string& key) {
    const Tensor& t = m.get(key);
    const size_t n = (size_t)t.numel();
    std::vector<float> out(n);
    if (t.dtype == F32) {
        QTTS_HIP(hipMemcpy(out.data(), t.buf.p, n * 4, hipMemcpyDeviceToHost));
    } else {
        std::vector<uint16_t> tmp(n);
        QTTS_HIP(hipMemcpy(tmp.data(), t.buf.p, n * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; ++i) out[i] = bf16_to_f32(tmp[i]);
    }
    return out;
}

inline void upload(DevBuf& d, const std::vector<float>& v, int dtype) {
    if (dtype == F32) {
        d.alloc(v.size() * 4);
        QTTS_HIP(hipMemcpy(d.p, v.data(), v.size() * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<uint16_t> tmp(v.size());
        for (size_t i = 0; i < v.size(); ++i) tmp[i] = f32_to_bf16(v[i]);
        d.alloc(tmp.size() * 2);
        QTTS_HIP(hipMemcpy(d.p, tmp.data(), tmp.size() * 2, hipMemcpyHostToDevice));
    }
}

struct GemmW {
    DevBuf w, bias;  // w: [N][K] model dtype (in fragment order when `packed`); bias: [CoutP] f32 (zeros when absent)
    bool packed = false;
    int N = 0, K = 0, taps = 1, dil = 1, r = 1, CinP = 0, CoutP = 0;
    WMat wm() const { return WMat(w.p, packed); }
};

// Uploads a host [N][K] f32 matrix in the model dtype, re-laid into MFMA fragment order (kernels.hpp WMat) when the shape
// tiles (N % 16 == 0, rows of whole 128-byte segments) and packing is enabled.
inline void upload_matrix(GemmW& g, const std::vector<float>& v, int N, int K, int dtype) {
    const size_t es = dtype_size(dtype);
    const size_t row_bytes = (size_t)K * es;
    QTTS_CHECK(v.size() == (size_t)N * K, ERR_SHAPE, "upload_matrix: size mismatch");
    std::vector<uint8_t> rm((size_t)N * row_bytes);
    if (dtype == F32) {
        std::memcpy(rm.data(), v.data(), rm.size());
    } else {
        uint16_t* o = reinterpret_cast<uint16_t*>(rm.data());
        for (size_t i = 0; i < v.size(); ++i) o[i] = f32_to_bf16(v[i]);
    }
    g.packed = pack_weights_enabled() && N % 16 == 0 && row_bytes % 128 == 0;
    g.w.alloc(rm.size());
    if (!g.packed) {
        QTTS_HIP(hipMemcpy(g.w.p, rm.data(), rm.size(), hipMemcpyHostToDevice));
        return;
    }
    std::vector<uint8_t> pk(rm.size());
    const size_t ppr = row_bytes / 16, nC = ppr / 8;
    for (size_t n = 0; n < (size_t)N; ++n)
        for (size_t q = 0; q < ppr; ++q) {
            const size_t c = q >> 3, pp = q & 7;
            const size_t d = ((((n >> 4) * nC + c) * 2 + (pp & 1)) * 64) + (n & 15) + 16 * (pp >> 1);
            std::memcpy(pk.data() + d * 16, rm.data() + (n * ppr + q) * 16, 16);
        }
    QTTS_HIP(hipMemcpy(g.w.p, pk.data(), pk.size(), hipMemcpyHostToDevice));
}

struct Snake {
    DevBuf a, den;  // exp(alpha), exp(beta) + 1e-9
};

inline void load_bias(const Model& m, const std::string& key, int CoutP, GemmW& g) {
    std::vector<float> b((size_t)CoutP, 0.f);
    if (!key.empty() && m.has(key)) {
        const std::vector<float> src = host_f32(m, key);
        QTTS_CHECK((int)src.size() <= CoutP, ERR_SHAPE, "bias longer than its layer: " + key);
        std::copy(src.begin(), src.end(), b.begin());
    }
    upload(g.bias, b, F32);
}

// Conv1d weight (Cout, Cin, k) (candle layout, causal_conv.rs:36-60) -> W'[co][tap * CinP + ci]
inline void conv_weight(const Model& m, const std::string& prefix, int dil, GemmW& g) {
    const Tensor& t = m.get(prefix + ".weight");
    QTTS_CHECK(t.shape.size() == 3, ERR_SHAPE, "conv weight must be rank 3: " + prefix);
    const int Cout = (int)t.shape[0], Cin = (int)t.shape[1], k = (int)t.shape[2];
    const std::vector<float> w = host_f32(m, prefix + ".weight");
    g.CinP = pad64(Cin);
    g.CoutP = pad64(Cout);
    g.taps = k;
    g.dil = dil;
    g.r = 1;
    g.N = g.CoutP;
    g.K = k * g.CinP;
    std::vector<float> out((size_t)g.N * g.K, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci)
            for (int j = 0; j < k; ++j) out[(size_t)co * g.K + (size_t)j * g.CinP + ci] = w[((size_t)co * Cin + ci) * k + j];
    upload_matrix(g, out, g.N, g.K, m.dtype);
    load_bias(m, prefix + ".bias", g.CoutP, g);
}

// ConvTranspose1d weight (Cin, Cout, k) with stride r, trimmed by k - r on both ends (causal_conv.rs:107-176).
//   k == 2r: out[t' r + j] = x[t'] W[:, :, j + r] + x[t' + 1] W[:, :, j]          (two taps, T - 1 input rows)
//   k == r:  out[t' r + j] = x[t'] W[:, :, j]                                      (one tap)
// Either way a GEMM with N = r * CoutP: W'[j * CoutP + co][tap * CinP + ci].
inline void convT_weight(const Model& m, const std::string& prefix, int r, GemmW& g) {
    const Tensor& t = m.get(prefix + ".weight");
    QTTS_CHECK(t.shape.size() == 3, ERR_SHAPE, "transposed conv weight must be rank 3: " + prefix);
    const int Cin = (int)t.shape[0], Cout = (int)t.shape[1], k = (int)t.shape[2];
    QTTS_CHECK(k == r || k == 2 * r, ERR_SHAPE, "transposed conv: only kernel == stride or kernel == 2 * stride is supported");
    const std::vector<float> w = host_f32(m, prefix + ".weight");
    g.CinP = pad64(Cin);
    g.CoutP = pad64(Cout);
    g.taps = k / r;
    g.dil = 1;
    g.r = r;
    g.N = r * g.CoutP;
    g.K = g.taps * g.CinP;
    std::vector<float> out((size_t)g.N * g.K, 0.f);
    for (int ci = 0; ci < Cin; ++ci)
        for (int co = 0; co < Cout; ++co)
            for (int j = 0; j < r; ++j) {
                const size_t row = (size_t)j * g.CoutP + co;
                if (g.taps == 2) {
                    out[row * g.K + ci] = w[((size_t)ci * Cout + co) * k + j + r];
                    out[row * g.K + g.CinP + ci] = w[((size_t)ci * Cout + co) * k + j];
                } else {
                    out[row * g.K + ci] = w[((size_t)ci * Cout + co) * k + j];
                }
            }
    upload_matrix(g, out, g.N, g.K, m.dtype);
    load_bias(m, prefix + ".bias", g.CoutP, g);
}

// Linear weight (out, in) -> padded [outP][inP]
inline void linear_weight(const Model& m, const std::string& prefix, GemmW& g) {
    const Tensor& t = m.get(prefix + ".weight");
    QTTS_CHECK(t.shape.size() == 2, ERR_SHAPE, "linear weight must be rank 2: " + prefix);
    const int Cout = (int)t.shape[0], Cin = (int)t.shape[1];
    const std::vector<float> w = host_f32(m, prefix + ".weight");
    g.CinP = pad64(Cin);
    g.CoutP = pad64(Cout);
    g.N = g.CoutP;
    g.K = g.CinP;
    std::vector<float> out((size_t)g.N * g.K, 0.f);
    for (int co = 0; co < Cout; ++co)
        for (int ci = 0; ci < Cin; ++ci) out[(size_t)co * g.K + ci] = w[(size_t)co * Cin + ci];
    upload_matrix(g, out, g.N, g.K, m.dtype);
    load_bias(m, prefix + ".bias", g.CoutP, g);
}

inline void padded_vec(const Model& m, const std::string& key, int CP, float fill, DevBuf& d) {
    std::vector<float> v = host_f32(m, key);
    QTTS_CHECK((int)v.size() <= CP, ERR_SHAPE, "vector longer than its layer: " + key);
    v.resize((size_t)CP, fill);
    upload(d, v, F32);
}

inline void load_snake(const Model& m, const std::string& prefix, int CP, Snake& s) {
    std::vector<float> a = host_f32(m, prefix + "alpha"), b = host_f32(m, prefix + "beta");
    QTTS_CHECK(a.size() == b.size() && (int)a.size() <= CP, ERR_SHAPE, "snake parameters: bad shape " + prefix);
    for (auto& v : a) v = std::exp(v);
    for (auto& v : b) v = std::exp(v) + (float)1e-9;
    a.resize((size_t)CP, 1.f);
    b.resize((size_t)CP, 1.f);
    upload(s.a, a, F32);
    upload(s.den, b, F32);
}


}  // namespace qtts
