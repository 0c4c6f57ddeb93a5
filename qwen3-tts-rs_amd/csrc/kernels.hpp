// Launchers of the gfx950 kernels.  Every launcher validates the shapes its kernel and grid assume
// before launching (an out-of-bounds access on this hardware can reset the whole node).
#pragma once
#include "common.hpp"

#include <vector>

namespace qtts {

// One token in flight: which KV-cache row it belongs to, its position (RoPE + cache slot) and the
// key range [ctx_start, ctx_start + ctx_len) it attends to.
struct Tok {
    int row, pos, ctx_start, ctx_len;
};

// Paged KV cache of one transformer stack.  A page holds `page_tokens` positions of every kv head of
// one layer: [layer][page][k|v][kv_head][slot][d].  page_table[row * max_pages + i] = physical page.
struct KVView {
    void* base;
    size_t layer_stride;  // elements
    int page_tokens, nkv, d, max_pages;
    const int* page_table;  // null: row r owns pages [r * max_pages, (r + 1) * max_pages)
    int page_mask = 0x7fffffff;  // ring caches: max_pages is a power of two and page_mask = max_pages - 1 (positions wrap)
};

// ---- k_gemm.hip
// A GEMM weight matrix W[N][K] (model dtype).  Row-major as loaded, or "packed" by launch_pack_weight into MFMA fragment
// order: in 16-byte pieces, piece pp (0..7) of the c-th 128-byte segment of row n lives at piece index
//   (((n / 16) * nC + c) * 2 + (pp & 1)) * 64 + (n % 16) + 16 * (pp >> 1),      nC = K * elem_size / 128,
// so that a wave's fragment load of a 16-row tile is one contiguous KiB.  N % 16 == 0 and K * elem_size % 128 == 0.
struct WMat {
    const void* p = nullptr;
    bool packed = false;
    WMat() = default;
    WMat(const void* q, bool pk = false) : p(q), packed(pk) {}
    explicit operator bool() const { return p != nullptr; }
};
void launch_pack_weight(hipStream_t st, const void* src, void* dst, int N, long row_bytes);

struct GemmPlan {
    int KS, kind, NT, NCH;  // kind 0: in-block split-K kernel, 1: LDS-shared-X kernel
    int MT;                 // kind 1: 16-row tiles per block (the rows are cut into Mpad / (16 MT) slices)
};
constexpr int GEMM_MAX_KS = 16;  // most split-K slabs a plan may ask for; the slab workspaces are sized for it
int gemm_pad_rows(int M);
GemmPlan gemm_plan(int M, int N, int K, int dtype);
int gemm_pick_ks(int M, int N, int K, int dtype);
// out is [KS][gemm_pad_rows(M)][N] f32 partial sums; X has gemm_pad_rows(M) readable rows of ldx elements.
// x_lo != 0 (bf16 only): X carries a remainder plane x_lo elements behind it and both planes are multiplied, so that the
// activations enter the product with about 16 significant bits (the weights stay plain bf16).
void launch_gemm(hipStream_t st, int dtype, WMat W, const void* X, float* out, int M, int N, int K, int KS, int ldx,
                 long x_lo = 0);
// Large-M GEMM with convolution addressing: out[m][n] = sum_{tap, c} X[m*ldx + tap*tap_stride + c] * W[n][tap*C + c], C = K / taps.
// out is [ceil64(M)][N] f32; X must be readable for ceil64(M) rows plus the tap offsets.
void launch_gemm_rows(hipStream_t st, int dtype, WMat W, const void* X, float* out, long M, int N, int K, int ldx, int taps,
                      long tap_stride, long x_lo = 0);

// ---- decode kernel choice for one stream alone on the GPU or several in flight (k_gemm_fused.hip)
// With several decode streams on one GPU (sessions) the kernels queue for each other's CUs, and what counts is a kernel's
// throughput next to copies of itself, not its latency alone.  Measured with tools/gemm_harness on one and on three streams
// (us per launch overall, 64 rows): talker q|k|v 4096 x 2048 - whole-K tile 10.1 / 8.75, K-streaming kernel 12.8 / 5.16;
// code-predictor gate|up 6144 x 1024 - 7.1 / 3.82 against 7.8 / 3.08; k_split at K = 3072 with six X chunks per request
// 6.0 / 3.78, with three 6.2 / 2.75.  In the whole three-session job: 1832 audio-s/s (no codec) with the kernels picked for one
// stream, 2000 with the K-streaming kernel wherever it fits and three chunks per k_split request.
// The choice belongs to a weight set (a model and its sessions) and is made ONCE, by whichever comes first: opening a session
// (several streams) or the first generate call (one stream); QTTS_TUNE=alone / shared forces it for every model.  It never
// changes afterwards, so a request's tokens do not depend on what else is open in the process.  Results of bf16 runs differ
// between the two choices by summation order only (within the same bounds against a CPU evaluation); f32 runs do not use
// these kernels.
enum { TUNE_UNSET = -1, TUNE_ALONE = 0, TUNE_SHARED = 1 };
int tuning_forced();  // QTTS_TUNE: TUNE_ALONE / TUNE_SHARED, or TUNE_UNSET when the variable does not force a choice

// ---- k_gemm_fused.hip: decode GEMMs that own finished outputs and apply the consumer's element-wise work (bf16, hidden <= 1024)
enum { FUSED_PLAIN = 0, FUSED_SWIGLU = 1, FUSED_RESID = 2 };
struct FusedEpi {
    int mode;
    // full_k (PLAIN / SWIGLU): finished sums are multiplied by 1 / sqrt(sum_t ss_in[t][row] / H_in + eps) (no scale when ss_in is null)
    const float* ss_in;   // [ntiles_in][ss_ld]
    long ss_ld;           // row stride of ss_in / ss_out
    int ntiles_in, H_in;
    float eps;
    const float* bias;    // [N] or null (PLAIN, RESID)
    float* out_f32;       // PLAIN: [rows][ld_out]
    int ld_out;
    void* act;            // SWIGLU: [rows][N / 2] bf16 (+ remainder plane act_lo elements behind, 0 = none)
    long act_lo;
    // k_split (RESID): x = [resid +] sum + bias ; xb = bf16(x * nw) (+ remainder plane) ; ss_out[tile][row] = sum over the tile's 16 columns of x^2
    const float* resid;   // [rows][N] f32 or null (may alias x_out)
    float* x_out;         // [rows][N] f32
    void* xb_out;         // [rows][N] bf16
    long xb_lo;
    const float* nw;      // [N] norm weight of the consumer
    float* ss_out;        // [N / 16][ss_ld]
};
bool fused_gemm_supported(int dtype, int hidden);
int fused_rows(int M);           // rows the kernels read and write for M tokens (16, 32 or a multiple of 64)
int fused_resid_tiles(int N);    // tiles a RESID launch writes to ss_out
// W rows for SWIGLU: every 16 rows = [4 gate, 4 up, 4 gate, 4 up] of eight consecutive channels
// full_k takes K <= 2048 as long as a 16-row X tile (both planes when x_lo != 0) fits LDS: full_k_fits
bool full_k_fits(int K, bool lo);
void launch_gemm_full_k(hipStream_t st, WMat W, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e);
// same contract as full_k for 64-row multiples and K = 1024 or 2048: K is walked in double-buffered phases inside the block
bool gemm_stream_fits(int M, int N, int K, bool lo);
bool gemm_stream_preferred(int M, int N, int K, bool lo, bool shared);  // it fits and is the faster kernel for the shape (shared: several decode streams in flight)
void launch_gemm_stream(hipStream_t st, WMat W, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e);
void launch_gemm_k_split(hipStream_t st, WMat W, const void* X, int ldx, long x_lo, int M, int N, int K, const FusedEpi& e, bool shared = false);

// ---- k_gemm_frag.hip: 64-row decode GEMM on activations stored in MFMA fragment order ("packed": the 16-byte piece (row tile i,
// chunk c, half h, lane) of a [64][K] bf16 matrix at piece index ((i K / 64 + c) 2 + h) 64 + lane); no LDS staging of X.  Same
// epilogue contract as the kernels above, except that activations it writes (xb_out, act) are packed too and always cover 64 rows.
// false = shape or weight layout not covered.
bool gemm_frag_covers(int N, int K, int mode);
bool launch_gemm_frag(hipStream_t st, WMat W, const void* Xp, int N, int K, const FusedEpi& e, int rows = 64);  // rows: 64 or 128 (packed rows of Xp)

// ---- k_bench.hip (diagnostics)
void launch_stream(hipStream_t st, const void* src, unsigned* sink, long bytes, int blocks, int pattern, int unroll, int ld_bytes);


// ---- k_elem.hip
// x = [resid] + [scale *] (sum_ks part[ks] + [bias]); resid_out = x (f32, may alias resid); xn = rmsnorm(x) * w -> dtype;
// xn_f32 (optional) receives the same normalised row in f32.
// `scale` (optional, per channel) multiplies (sum part + bias) before the residual add (LayerScale).
// Every `*_lo` argument below: element offset of the output's remainder plane (see launch_gemm), 0 = none.
void launch_resid_norm(hipStream_t st, int dtype, const float* resid, const float* part, int KS, long ks_stride,
                       const float* bias, const float* scale, const float* norm_w, float eps, float* resid_out, void* xn,
                       float* xn_f32, int M, int H, long xn_lo = 0);
// split-K combine + residual; xb = bf16(x * norm_w) (+ remainder plane xb_lo elements behind) and per-row sums of x^2 over slices of
// 256 columns -> ss_out[slice][ss_ld] for a consumer GEMM that applies 1/rms (FusedEpi::ss_in).  Returns the number of slices.
int launch_resid_partial(hipStream_t st, const float* resid, const float* part, int KS, long ks_stride, const float* bias,
                         const float* norm_w, float* resid_out, void* xb, long xb_lo, float* ss_out, long ss_ld, int M, int H);
// act = silu(sum part[..][0:I]) * sum part[..][I:2I] -> dtype   (candle_nn::ops::swiglu)
void launch_swiglu(hipStream_t st, int dtype, const float* part, int KS, long ks_stride, void* act, int M, int I, long act_lo = 0);
// out = act(sum_ks part + bias): act 0 none, 1 silu, 2 erf-GELU; out_dtype F32/BF16 independent of the weight dtype
void launch_bias_act(hipStream_t st, const float* part, int KS, long ks_stride, const float* bias, int act, void* out,
                     int out_dtype, int M, int N, long out_lo = 0);
// dst[i][:] = table[idx[i]][:]  (table in `tdtype`, dst in `ddtype`)
void launch_gather_rows(hipStream_t st, const void* table, int tdtype, const int* idx, void* dst, int ddtype, int M, int H,
                        long dst_ld);
// prompt row = text_rows[text_idx] + codec_emb[codec_id] + extra[extra_idx]  (each optional: index < 0 skips)
void launch_assemble(hipStream_t st, const float* text_rows, const int* text_idx, const void* codec_emb, int edtype,
                     const int* codec_id, const float* extra, const int* extra_idx, float* out, int M, int H);
// next talker input: codec_emb[c0] + sum_g cp_emb[g-1][c_g] + (step < Tt ? trailing[step] : tts_pad), then the
// layer-0 input RMSNorm.  codes points at this step's [B][G] u32 block through (codes_base + *step * step_stride).
struct EmbedSumArgs {
    const void* codec_emb;          // [V][H]
    const void* const* cp_emb;      // device array of G-1 table pointers, each [Vc][H]
    int edtype;
    const unsigned* codes_base;     // [B][max_steps][G]
    long row_stride;                // max_steps * G
    const int* step;                // [B]: every row's frame index
    int max_steps;                  // frames a row's slice of codes holds (a retired row re-reads its last frame)
    const float* trailing;          // [B][Tt_max][H] f32
    const int* trailing_len;        // [B]
    int Tt_max;
    const float* tts_pad;           // [H] f32
    int G, H, B;
};
void launch_embed_sum_norm(hipStream_t st, int dtype, const EmbedSumArgs& a, const float* norm_w, float eps, float* resid_out,
                           void* xn, long xn_lo = 0);
// fused decode stacks: x (f32 rows) -> xb = T(x * nw) (+ remainder plane), ss_out[row] = sum x^2 (one norm partial tile)
// x_out (optional) receives a copy of x
void launch_rows_prep(hipStream_t st, int dtype, const float* x, const float* nw, float* x_out, void* xb, long xb_lo, float* ss_out, int M, int H);
// ICL prompt rows: dst[rows[i]][:] = codec_emb[codes[i][0]] + sum_g cp_emb[g-1][codes[i][g]]  (f32)
void launch_sum_code_rows(hipStream_t st, int edtype, const void* codec_emb, const void* const* cp_emb, const int* codes, const int* rows,
                          int n, int G, int H, float* dst);
void launch_cast_rows(hipStream_t st, const float* src, void* dst, int ddtype, long n);
// decode-step bookkeeping (device-side so the frame can live in a hipGraph)
// toks[b] = {b, row_len[b], window start, ctx}; row_len[b] += 1 (rows with active[b] == 0 stay where they are)
void launch_make_decode_toks(hipStream_t st, int* row_len, Tok* toks, int B, int window, const int* active);
// step[b] += 1 for active rows.  retire != 0: a row switches itself off once it has produced max_steps frames or the frame with
// its first EOS (eos_step[b], -1 = none yet) - what a batch of one would have generated
// row_max (optional, B ints): each row's own frame limit (the smaller of it and max_steps retires the row)
void launch_advance_step(hipStream_t st, int* step, int* active, const int* eos_step, int B, int max_steps, int retire,
                         const int* row_max = nullptr);
void launch_copy_rows(hipStream_t st, const void* src, const int* sidx, void* dst, const int* didx, int dtype, int n, int H);

// ---- k_attn.hip
struct QkvPostArgs {
    const float* part;   // [KS][Mpad][(nh + 2 nkv) d]
    int KS;
    long ks_stride;
    const float* bias;   // optional [(nh + 2 nkv) d]
    const float* q_norm; // [d] or null (no per-head norm)
    const float* k_norm;
    float eps;
    const float* cos;    // [max_pos][d/2]
    const float* sin;
    int rope_mode;       // 0: candle rope (pair i, i+d/2 with half-dim tables)
    const Tok* toks;
    int M, nh, nkv, d, layer;
    float* q_out;        // [Mpad][nh d] f32
    int uni_len = 0;     // decode launches only: > 0 = every token m is Tok{m, uni_len - 1, 0, uni_len}; toks is not read
    int small = 0;       // 1 = several decode streams share the GPU: launches with uni_len <= 16 may take the one-wave-per-(token, kv head)
                         // kernel and growing contexts keep four waves per block (k_attn.hip)
    int out_packed = 0;  // decode launches, bf16 only: > 0 = the attention output is a packed [64][out_packed] matrix (device_utils.hpp packed_elem)
};
void launch_qkv_post(hipStream_t st, int dtype, const QkvPostArgs& a, const KVView& kv);
// softmax(q k^T * scale over the token's key range) v  -> out [Mpad][nh d] in dtype
void launch_attention(hipStream_t st, int dtype, const float* q, const Tok* toks, const KVView& kv, int layer, int M, int nh,
                      float scale, void* out, long out_lo = 0);

// qkv_post + attention in one launch for decode steps: every token's newest key is its own position and no other token
// of the launch attends to it
void launch_attention_decode(hipStream_t st, int dtype, const QkvPostArgs& a, const KVView& kv, float scale, void* out, long out_lo = 0);

// ---- k_attn_tile.hip: attention on the matrix cores for launches whose tokens come in runs (prefill chunks, the codec's
// banded transformer).  A tile = up to ATTN_TILE_TOKENS tokens m0 .. m0 + count - 1 of the launch that belong to one cache row at
// consecutive positions; [k0, k1) covers the key ranges of all of them (each token still sees only its own range).
constexpr int ATTN_TILE_TOKENS = 64;
struct AttnTile {
    int m0, count, row, k0, k1;
};
// cuts the M tokens of a launch into tiles (any token order is legal: a token that continues no run starts a tile of its own)
void build_attn_tiles(const Tok* toks, int M, std::vector<AttnTile>& tiles);
bool attention_tiles_supported(int d, int nh, int nkv);
// same contract as launch_attention; the keys and values a tile sees are fetched once per tile and shared through LDS
void launch_attention_tiles(hipStream_t st, int dtype, const float* q, const Tok* toks, const AttnTile* tiles, int ntiles, const KVView& kv,
                            int layer, int nh, float scale, void* out, long out_lo = 0);

// ---- k_sample.hip
struct SampleArgs {
    const float* logits;    // [B][V]
    int V, B;
    float temperature;      // 1/T applied as a multiply
    int top_k;
    float top_p;
    int do_sample;
    float rep_penalty;      // 1.0 disables
    int* hist_count;        // [B][V] occurrences so far (null when rep_penalty == 1)
    int suppress_lo, suppress_hi, eos_id;  // eos_id < 0: none
    int min_new_tokens;
    const int* step;        // [B]: every row's frame index (== its history length)
    const int* active;      // [B] or null: rows with 0 draw on their last frame index and write nothing
    int max_steps;          // frames a row's slice of codes / uniforms holds
    const float* uniforms;  // [B][max_steps][G]
    unsigned* codes_base;   // [B][max_steps][G]
    long row_stride;
    int G, g;               // which codebook column this draw fills
    int* eos_step;          // [B], -1 until the row's first EOS (only for g == 0)
    // optional: copy next_table[token] (H elements, bf16 or f32) to next_dst row b (same dtype, or f32 from a bf16 table)
    const void* next_table;
    void* next_dst;
    int next_H, next_table_bf16, next_dst_bf16;
    // optional: gather two f32 rows by the drawn token (tabulated start of the next code-predictor pass): memo_x[token] ->
    // memo_x_dst row b (memo_Hx floats), memo_q[token] -> memo_q_dst row b (memo_Nq floats); both widths multiples of 4
    const float* memo_x;
    float* memo_x_dst;
    const float* memo_q;
    float* memo_q_dst;
    int memo_Hx, memo_Nq;
};
void launch_sample(hipStream_t st, const SampleArgs& a);

// ---- k_codec.hip
struct ConvEpi {
    const float* acc;        // [Mpad][N] f32; row m = u * S_in + t'
    long M;                  // rows to visit
    int N, r, Cout;          // N = r * Cout (r > 1: transposed conv folded into N)
    const float* bias;       // [Cout] or null
    const float* bias_u;     // [U][Cout] or null: extra per-strip bias
    int Cvalid;              // 0: write all Cout columns; otherwise only columns < Cvalid (multiple of 4)
    const float* scale;      // [Cout] or null (ConvNeXt gamma)
    int act;                 // 0 none, 1 tanh-GELU, 2 ReLU, 3 tanh(ReLU), 4 ELU
    int act2;                // activation of out2: 0 SnakeBeta (alpha_exp / beta_den), 1 ELU
    const int* lens;         // [U] valid GEMM rows per strip
    long S_in, S_out;        // strip lengths (rows) of the GEMM row space / of the output tensor
    int P_out, ld_out;       // output row = u * S_out + P_out + t' * r + j ; ld_out = channel stride of out / out2 / resid
    const void* resid;       // optional, same layout as out (model dtype)
    const float* resid_f32;  // optional, same layout as out (f32)
    void* out;               // model dtype, optional
    float* out_f32;          // optional f32 copy
    void* out2;              // optional SnakeBeta(out)
    const float* alpha_exp;  // exp(alpha)
    const float* beta_den;   // exp(beta) + 1e-9
    // bf16 runs: element offsets of the remainder planes of resid / out / out2 (see launch_gemm), 0 = single plane
    long resid_lo, out_lo, out2_lo;
};
// The element-wise tail moves four consecutive channels per access: channel strides and plane offsets are multiples of four
// elements (the launchers check it; the buffers themselves come from 256-byte aligned allocations).
inline bool conv_epi_vector_ok(const ConvEpi& e) {
    return e.ld_out % 4 == 0 && e.out_lo % 4 == 0 && e.out2_lo % 4 == 0 && e.resid_lo % 4 == 0;
}

void launch_conv_epilogue(hipStream_t st, int dtype, const ConvEpi& e);
// GEMM + epilogue in one launch (k_gemm.hip): rows e.M, columns e.N; e.acc is not used
// x_lo != 0: X carries a remainder plane (tap-addressed kernel only)
void launch_gemm_rows_epi(hipStream_t st, int dtype, WMat W, const void* X, int K, int ldx, int taps, long tap_stride,
                          const ConvEpi& e, long x_lo = 0);
// k_conv.hip: same contract with the input tile held in LDS; false = shape not covered (use launch_gemm_rows_epi).
// X must be readable for CONV_HALO_SLACK rows past e.M: the last block's tile starts below e.M and spans its rows, the halo of
// the taps and up to 8 rows of DMA rounding; a shape whose tile would reach further is refused (false).
constexpr long CONV_HALO_SLACK = 448;
// x_lo != 0 (bf16): X carries a remainder plane (see launch_gemm); covered by the wave-grid tile kernel only
bool launch_conv_halo(hipStream_t st, int dtype, WMat W, const void* X, int K, int ldx, int taps, long tap_stride, const ConvEpi& e,
                      long x_lo = 0);
bool conv_unit_supported(int dtype, int C, int taps, int dil);
// The fused unit at 96 channels without padding (k_conv.hip): weight layout and launch
void conv_unit96_pack(const float* w, int taps, uint16_t* out, uint16_t (*to_bf16)(float));
// the decoder's output conv (96 -> 1, seven taps, clamp) when it runs inside the last unit's kernel
struct Unit96Final {
    const float* w = nullptr;  // [7][96]
    float bias = 0.f;
    float* wav = nullptr;      // strip u's samples start at wav + u * wav_stride
    long wav_stride = 0;
};
bool launch_conv_unit96(hipStream_t st, const void* W1, const void* W2, int dil, const float* bias1, const float* alpha2, const float* den2,
                        const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1, const Unit96Final* fin = nullptr);
// Two consecutive 96-channel units (dilations 1 and 3) in one launch.  The first unit's arguments as for launch_conv_unit96
// (e.bias = ITS 1x1 bias; e.resid_f32 / Xf = the stream it reads); e's outputs are the PAIR's outputs; the *b arguments are the
// second unit's weights, biases (bias2b: its 1x1 bias), SnakeBeta of its 7-tap conv and its input activation.
bool launch_conv_unit96_pair(hipStream_t st, const void* W1, const void* W2, int dilA, int dilB, const float* bias1, const float* alpha2,
                             const float* den2, const ConvEpi& e, const float* Xf, const float* alpha1, const float* den1, const void* W1b,
                             const void* W2b, const float* bias1b, const float* alpha2b, const float* den2b, const float* bias2b,
                             const float* alpha1b, const float* den1b);
// Xf (bf16 runs, X null): the unit makes its activated input tile itself from the f32 stream Xf (same addressing as X) with the
// input activation (alpha1, den1) - no activated copy is read, and the producer need not write one.
bool launch_conv_unit(hipStream_t st, int dtype, WMat W1, WMat W2, const void* X, int C, int taps, int dil, const float* bias1,
                      const float* alpha2, const float* den2, const ConvEpi& e, const float* Xf = nullptr, const float* alpha1 = nullptr,
                      const float* den1 = nullptr);
void launch_rvq_gather(hipStream_t st, int dtype, const float* table, const unsigned* codes, const int* rows, const int* starts,
                       const int* lens, int U, int Tc, int Q, int csize, int Dc, int T_max, long S, int P, void* out, int ld, long out_lo = 0);
void launch_dwconv_ln(hipStream_t st, int dtype, const void* x, const float* w, const float* b, const float* ln_w, const float* ln_b,
                      const int* lens, int U, int Tmax, long S, int P, int C, void* out, long x_lo = 0);
void launch_final_conv(hipStream_t st, int dtype, const void* x, const float* w, float bias, const int* nvalid, int U, long Tmax, long S,
                       int P, int C, float* wav, long wav_stride);

// ---- k_speaker.hip
void launch_mel(hipStream_t st, int dtype, const float* audio, const int* n_samples, int B, int max_samples, int max_frames,
                const float* window, const float* twiddle, const float* fbT, float* mel_out, long frames_stride, void* strips, long S,
                int P);
void launch_reflect_fill(hipStream_t st, int dtype, void* x, const int* lens, int U, long S, int P, int pad, int C);
void launch_res2net_prep(hipStream_t st, int dtype, const void* a, const void* prev, const int* lens, int U, int Tmax, long S, int P,
                         int pad, int ld, int off, int cs, void* tmp);
void launch_slice_copy(hipStream_t st, int dtype, const void* a, const int* lens, int U, int Tmax, long S, int P, int ld, int cs, void* out);
void launch_time_stats(hipStream_t st, int dtype, const void* x, int ldx, const float* logits, int ldl, const float* lbias, const int* lens,
                       int U, long S, int P, int Pl, int C, int mode, float* out);  // Pl: leading rows of the logits' row space
void launch_small_fc(hipStream_t st, const float* W, int ldw, const float* b, const float* x, int ldx, int U, int N, int K, int act, float* y,
                     int ldy);
void launch_se_apply(hipStream_t st, int dtype, const void* x, int ldx, const float* gate, const void* resid, int ldr, const int* lens, int U,
                     int Tmax, long S, int P, int C, void* out, int ldo);

// ---- k_encoder.hip (codec encoder)
void launch_enc_init_conv(hipStream_t st, int dtype, const float* audio, const int* n_samples, int U, long max_samples, const float* w,
                          const float* bias, long S, int P, void* out_raw, void* out_elu);
// x = resid + [scale *] sum_ks part ; resid_out = x ; xn = LayerNorm(x) * ln_w + ln_b (skipped when ln_w is null)
void launch_resid_layernorm(hipStream_t st, int dtype, const float* resid, const float* part, int KS, long ks_stride, const float* scale,
                            const float* ln_w, const float* ln_b, float eps, float* resid_out, void* xn, int M, int H);
// valid rows of f32 strips -> model-dtype strips (other rows untouched)
void launch_rows_to_strips(hipStream_t st, int dtype, const float* src, long S_src, int P_src, const int* lens, int U, int Tmax, int C,
                           void* dst, long S_dst, int P_dst);
// nearest codebook entry per row: idx = argmin_r (c2[r] - dots[m][r]) (first minimum); residual[m] -= emb[idx];
// residual_t = T(residual); codes[(u * T_cap + t) * Q + q] = idx for valid rows (m = u * S + t, t < lens[u])
void launch_vq_argmin(hipStream_t st, int dtype, const float* dots, const float* c2, const float* emb, int bins, int cd, float* residual,
                      void* residual_t, const int* lens, long S, int U, unsigned* codes, int T_cap, int Q, int q);

}  // namespace qtts
