// ConditionalGeneration::generate on the GPU (qwen_tts/src/nn/generation.rs:815-1194) for a batch of
// independent utterances: prompt embedding assembly (:510-799), talker prefill, then one hipGraph
// replay per generated frame covering the talker decode step, cb0 sampling, the G-1 code-predictor
// passes (nn/code_predictor.rs:433-531) and the next-input embedding sum.  No host round trip
// happens inside a frame; the host only polls the per-row EOS steps every few frames.
//
// Two drivers share every piece: the reference's batch (n utterances = n rows, all rows run until the last one is done),
// and continuous batching (n utterances through a fixed number of row slots: a row that has produced its EOS frame or
// max_new_tokens frames is retired at the next poll and its slot is prefilled with the next queued utterance while the other
// rows keep decoding; every utterance gets exactly the frames a batch of one would have produced for it).
#include "generate.hpp"
#include "cp_pass.hpp"
#include "runner.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>

namespace qtts {

namespace {

constexpr int PAGE_TOKENS = 64;      // talker KV page (positions per page)
constexpr int CP_PAGE_TOKENS = 64;   // one page holds a whole code-predictor context (<= groups + 1)
static const int PREFILL_CHUNK = getenv("QTTS_PREFILL_CHUNK") ? std::max(64, atoi(getenv("QTTS_PREFILL_CHUNK"))) : 4096;   // prompt tokens per prefill pass (34 ms at 1024, 29 ms at 4096 for 64 prompts)
constexpr int EOS_POLL = 8;          // frames between host checks of the EOS flags

struct HostMT {  // MT19937 (rand_mt `Mt::new(seed as u32)`), 24-bit uniforms (nn/mt_rng.rs:64-81)
    uint32_t mt[624];
    int idx = 624;
    explicit HostMT(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int i = 0; i < 624; ++i) {
                const uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
                mt[i] = mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    float uniform() { return (float)(next() & 0x00FFFFFFu) / 16777216.0f; }
};

std::string lower(const char* s) {
    std::string r = s ? s : "";
    for (auto& c : r) c = (char)std::tolower((unsigned char)c);
    return r;
}

}  // namespace

struct StackRun {
    DevBuf pool, table;
    KVView kv{};
    int layers = 0;
};

struct GenState {
    // activations (token-major rows)
    Workspace ws;                            // activation scratch shared by every layer
    DevBuf cp_in_t, cp_in_f, hid_scratch_f; // code-predictor inputs; rows [0,B) double as the talker's post-norm hidden
    DevBuf cp_qkv_in;                        // first-layer q|k|v rows gathered from the model's tables (CpMemo)
    DevBuf cp_pass_ts;
    DevBuf cp_pass;                          // flags and per-XCD activation buffers of the one-launch code-predictor pass
    DevBuf text_rows, extra, trailing, tts_pad_dummy;
    DevBuf ibuf;                             // int scratch: ids / descriptors
    DevBuf toks_prompt, tiles_prompt, toks_dec, toks_cp;
    DevBuf row_len, step, eos_step, active, row_max, hist, uniforms, codes, trailing_len, tts_pad;
    DevBuf trace_logits, trace_hidden, trace_cp;
    StackRun talker, cp;
    size_t cap_rows = 0;
    GemmTimer timer;                         // optional HIP-event timing of every GEMM launch
    DevBuf icl;                              // reference code ids + destination rows of ICL prompts
    hipGraphExec_t graph_exec = nullptr;
    ~GenState() {
        if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    }
};

// Host-side plan of the prompts of a group of requests (generation.rs:452-799): flattened position descriptors, the text ids
// whose projections the prompts and the trailing text need, speaker vectors / ICL rows, one Tok per prompt position.
// Request ids[i] occupies KV-cache row slots[i].
struct PromptPlan {
    std::vector<int> text_ids;                 // rows 0, 1, 2 of the projected-text buffer are tts_bos, tts_eos, tts_pad
    std::vector<int> p_text, p_codec, p_extra;  // flattened prompt position descriptors
    std::vector<Tok> p_toks;
    std::vector<int> row_P, trail_len;
    std::vector<std::vector<int>> trail_idx;
    std::vector<float> extra_rows;
    std::vector<int> icl_codes, icl_rows;      // reference frames whose embedding sums are computed on the device
};

static void plan_prompts(const ModelCfg& c, const qtts_request* reqs, const std::vector<int>& ids, const std::vector<int>& slots, PromptPlan& pl) {
    const int B = (int)ids.size(), H = c.talker.hidden, G = c.groups, V = c.vocab, Vc = c.cp_vocab;
    std::vector<int>& text_ids = pl.text_ids;
    std::vector<int>&p_text = pl.p_text, &p_codec = pl.p_codec, &p_extra = pl.p_extra;
    std::vector<Tok>& p_toks = pl.p_toks;
    std::vector<int>&row_P = pl.row_P, &trail_len = pl.trail_len;
    std::vector<std::vector<int>>& trail_idx = pl.trail_idx;
    std::vector<float>& extra_rows = pl.extra_rows;
    std::vector<int>&icl_codes = pl.icl_codes, &icl_rows = pl.icl_rows;
    text_ids = {c.tts_bos, c.tts_eos, c.tts_pad};
    row_P.assign((size_t)B, 0);
    trail_len.assign((size_t)B, 0);
    trail_idx.assign((size_t)B, {});
    auto add_text = [&](uint32_t id) {
        QTTS_CHECK((int)id < c.text_vocab, ERR_BAD_ARG, "generate: text id outside the text vocabulary");
        text_ids.push_back((int)id);
        return (int)text_ids.size() - 1;
    };
    for (int b = 0; b < B; ++b) {
        const qtts_request& r = reqs[ids[(size_t)b]];
        QTTS_CHECK(r.text_ids && r.n_text >= 4, ERR_BAD_ARG, "generate: text_ids needs at least 4 tokens");
        const bool icl = r.ref_codes != nullptr && r.ref_T > 0 && r.ref_text_ids != nullptr && r.n_ref_text > 0;
        QTTS_CHECK(r.ref_T >= 0 && (r.ref_codes == nullptr) == (r.ref_T == 0), ERR_BAD_ARG, "generate: ref_codes and ref_T disagree");
        // the reference slices ref_text_ids[3..-2] and fails on a shorter transcript (generation.rs:629-640)
        QTTS_CHECK(!icl || r.n_ref_text >= 4, ERR_BAD_ARG, "generate: ICL reference text needs at least 4 tokens");
        // build_codec_prefill (generation.rs:452-498)
        std::string lang = lower(r.language ? r.language : "auto");
        const std::string spk = lower(r.speaker);
        if ((lang == "chinese" || lang == "auto") && !spk.empty()) {
            auto it = c.spk_dialect.find(spk);
            if (it != c.spk_dialect.end()) lang = lower(it->second.c_str());
        }
        std::vector<int> ci;  // codec-side ids; -2 marks the external speaker vector
        auto lit = c.language_id.find(lang);
        if (lang != "auto" && lit != c.language_id.end()) ci = {c.think, c.think_bos, lit->second, c.think_eos};
        else ci = {c.nothink, c.think_bos, c.think_eos};
        int extra_row = -1;
        if (r.speaker_embed) {
            extra_row = (int)(extra_rows.size() / (size_t)H);
            extra_rows.insert(extra_rows.end(), r.speaker_embed, r.speaker_embed + H);
            ci.push_back(-2);
        } else if (!spk.empty()) {
            auto sit = c.spk_id.find(spk);
            if (sit != c.spk_id.end()) ci.push_back(sit->second);
        }
        ci.push_back(c.pad);
        ci.push_back(c.bos);
        const int L = (int)ci.size();
        const size_t first = p_text.size();
        auto push = [&](int t, int cd, int ex) { p_text.push_back(t); p_codec.push_back(cd); p_extra.push_back(ex); };
        auto codec_part = [&](int i, int& cd, int& ex) {
            if (ci[(size_t)i] == -2) { cd = -1; ex = extra_row; }
            else { cd = ci[(size_t)i]; ex = -1; }
        };
        for (int i = 0; i < r.n_instruct; ++i) push(add_text(r.instruct_ids[i]), -1, -1);
        for (int i = 0; i < 3; ++i) push(add_text(r.text_ids[i]), -1, -1);
        for (int i = 0; i < L - 1; ++i) {
            int cd, ex;
            codec_part(i, cd, ex);
            push(i < L - 2 ? 2 : 0, cd, ex);
        }
        if (icl) {
            // generate_icl_prompt (generation.rs:292-416, 629-696): text side = ref text (3..-2) || text (3..-5) || tts_eos,
            // codec side = bos || per-frame sums of the reference codes over all codebooks
            std::vector<int> te;
            const int rte = r.n_ref_text >= 2 ? r.n_ref_text - 2 : 0;
            for (int i = 3; i < (rte > 3 ? rte : 4); ++i) te.push_back(add_text(r.ref_text_ids[i]));
            const int tce = r.n_text >= 5 ? r.n_text - 5 : 0;
            for (int i = 3; i < (tce > 3 ? tce : 4); ++i) te.push_back(add_text(r.text_ids[i]));
            te.push_back(1);
            const int text_lens = (int)te.size(), codec_lens = r.ref_T + 1;
            const int icl_row0 = (int)(extra_rows.size() / (size_t)H);
            extra_rows.resize(extra_rows.size() + (size_t)r.ref_T * H, 0.f);
            for (int t = 0; t < r.ref_T; ++t) {
                for (int gi = 0; gi < G; ++gi) {
                    const uint32_t code = r.ref_codes[(size_t)t * G + gi];
                    QTTS_CHECK((int)code < (gi == 0 ? V : Vc), ERR_BAD_ARG, "generate: reference code id out of range");
                    icl_codes.push_back((int)code);
                }
                icl_rows.push_back(icl_row0 + t);
            }
            auto ce = [&](int i, int& cd, int& ex) {
                if (i == 0) { cd = c.bos; ex = -1; }
                else { cd = -1; ex = icl_row0 + i - 1; }
            };
            if (r.non_streaming) {
                for (int i = 0; i < text_lens; ++i) push(te[(size_t)i], c.pad, -1);
                for (int i = 0; i < codec_lens; ++i) { int cd, ex; ce(i, cd, ex); push(2, cd, ex); }
            } else if (text_lens > codec_lens) {
                for (int i = 0; i < codec_lens; ++i) { int cd, ex; ce(i, cd, ex); push(te[(size_t)i], cd, ex); }
                for (int i = codec_lens; i < text_lens; ++i) trail_idx[(size_t)b].push_back(te[(size_t)i]);
            } else {
                for (int i = 0; i < codec_lens; ++i) { int cd, ex; ce(i, cd, ex); push(i < text_lens ? te[(size_t)i] : 2, cd, ex); }
            }
        } else if (r.non_streaming && r.n_text > 8) {
            for (int i = 3; i < r.n_text - 5; ++i) push(add_text(r.text_ids[i]), c.pad, -1);
            push(1, c.pad, -1);
            push(2, c.bos, -1);
        } else {
            int cd, ex;
            codec_part(L - 1, cd, ex);
            push(add_text(r.text_ids[3]), cd, ex);
            if (r.n_text > 9) {
                for (int i = 4; i < r.n_text - 5; ++i) trail_idx[(size_t)b].push_back(add_text(r.text_ids[i]));
                trail_idx[(size_t)b].push_back(1);
            }
        }
        trail_len[(size_t)b] = (int)trail_idx[(size_t)b].size();
        row_P[(size_t)b] = (int)(p_text.size() - first);
        for (int p = 0; p < row_P[(size_t)b]; ++p) {
            Tok t;
            t.row = slots[(size_t)b];
            t.pos = p;
            t.ctx_start = c.talker.window > 0 ? std::max(0, p + 1 - c.talker.window) : 0;
            t.ctx_len = p + 1 - t.ctx_start;
            p_toks.push_back(t);
        }
    }
}

// slots > 0: continuous batching of n utterances through `slots` rows (see the top of the file); slots == 0: the reference's batch
static void generate_impl(Model& m, const qtts_request* reqs, int n, const qtts_sampling* sp, int max_new, qtts_codes_out* outs,
                          const qtts_gen_debug* dbg, int slots) {
    QTTS_CHECK(m.finalized && m.codec_head.p != nullptr, ERR_STATE, "generate: model has no talker weights or is not finalized");
    QTTS_CHECK(reqs && outs && sp && n >= 1 && (slots > 0 || n <= 1024) && n <= (1 << 20), ERR_BAD_ARG, "generate: bad batch");
    QTTS_CHECK(slots >= 0 && slots <= 1024, ERR_BAD_ARG, "generate: row slots out of range");
    QTTS_CHECK(max_new >= 1 && max_new <= 32768, ERR_BAD_ARG, "generate: max_new_tokens out of range");
    QTTS_HIP(hipSetDevice(m.ctx->device));
    const ModelCfg& c = m.cfg;
    const bool streaming = slots > 0;                 // continuous batching
    const int B = streaming ? std::min(slots, n) : n;  // rows of every launch
    const int H = c.talker.hidden, Ht = c.text_hidden, G = c.groups, V = c.vocab, Vc = c.cp_vocab, Hc = c.cp.hidden;
    if (!m.gen) m.gen.reset(new GenState());
    GenState& g = *m.gen;
    m.tuning->latch(TUNE_ALONE);  // first generate call of a weight set without sessions: kernels picked for one stream (kernels.hpp)
    Runner R(m, g.ws, &g.timer);
    hipStream_t st = R.st;
    const int dt = m.dtype;
    const size_t es = R.es;
    const bool use_graph = dbg ? dbg->use_graph != 0 : true;
    const bool split = split_activations(dt), split_cp = split_level(dt) >= 2;
    g.timer.enabled = dbg && dbg->profile_gemm && !use_graph;
    g.timer.used = 0;
    g.timer.bytes = 0;
    g.timer.shape.clear();
    const bool trace_l = dbg && dbg->logits, trace_h = dbg && dbg->hidden, trace_c = dbg && dbg->cp_logits;
    const bool tracing = trace_l || trace_h || trace_c;
    const bool shared_rng = dbg && dbg->shared_rng_stream != 0;
    QTTS_CHECK(!streaming || !shared_rng, ERR_BAD_ARG, "generate: one draw stream for all rows needs the plain batch");
    QTTS_CHECK(!streaming || !(tracing || (dbg && dbg->chunk_cb)), ERR_BAD_ARG, "generate: traces and the first-chunk hook need the plain batch");

    // a request's own frame limit applies to continuous batching only: the plain batch runs every row to the call's limit, as
    // the reference's batch does
    auto row_limit_of = [&](int u) {
        const int own = reqs[u].max_new_tokens;
        return streaming && own > 0 ? std::min(own, max_new) : max_new;
    };

    // ------------------------------------------------------------------ sizes over every request (the buffers the frame graph
    // points at are allocated once: prompt length and trailing-text length of the longest utterance)
    int Tt_max = 1, max_ctx = 0;
    {
        PromptPlan all;
        std::vector<int> ids((size_t)n), rows((size_t)n, 0);
        for (int i = 0; i < n; ++i) ids[(size_t)i] = i;
        plan_prompts(c, reqs, ids, rows, all);
        for (int i = 0; i < n; ++i) {
            Tt_max = std::max(Tt_max, all.trail_len[(size_t)i]);
            max_ctx = std::max(max_ctx, all.row_P[(size_t)i] + row_limit_of(i));
        }
    }

    // ------------------------------------------------------------------ workspace
    const int chunk_rows = std::max(PREFILL_CHUNK, gemm_pad_rows(2 * B));
    const size_t rows = (size_t)chunk_rows + 64;
    const int Wmax = std::max({H, Ht, Hc});
    const int QD = std::max({c.talker.nh * c.talker.d, c.cp.nh * c.cp.d, H, Hc});
    const int Imax = std::max({c.talker.inter, c.cp.inter, Ht, H, Hc});
    const int Nmax = std::max({(c.talker.nh + 2 * c.talker.nkv) * c.talker.d, (c.cp.nh + 2 * c.cp.nkv) * c.cp.d, 2 * c.talker.inter,
                               2 * c.cp.inter, V, Vc, H, Ht, Hc});
    g.ws.x.ensure(rows * Wmax * 4);
    g.ws.alloc_inputs(rows, (size_t)Wmax, (size_t)QD, (size_t)Imax, es, split);
    R.split(true);  // the runner's copy of the plane offsets follows the allocation
    // code predictor as five launches per layer (bf16, hidden <= 1024, single-plane activations)
    const bool fused_cp = Runner::stack_fusable(m, m.cp) && !split_cp && Vc % 64 == 0 && H % 64 == 0;
    if (fused_cp || dt == BF16) {  // (the talker's decode steps use ssB between the o projection and gate|up)
        g.ws.ss_ld = (long)fused_rows(2 * B) + 64;
        g.ws.ssA.ensure((size_t)(std::max(H, Hc) / 16 + 1) * g.ws.ss_ld * 4);
        g.ws.ssB.ensure((size_t)(std::max(H, Hc) / 16 + 1) * g.ws.ss_ld * 4);
    }
    // up to GEMM_MAX_KS split-K slabs for the small-M kernels (at most 512 rows); the large-M kernel writes one [rows][N] matrix
    g.ws.part.ensure(std::max(rows, (size_t)GEMM_MAX_KS * std::min<size_t>(rows, 512)) * (size_t)Nmax * 4);
    g.ws.q.ensure(rows * QD * 4);
    const size_t cp_rows = (size_t)gemm_pad_rows(2 * B) + 64;
    // code-predictor input rows: [0, B) the talker's post-norm hidden (split like every GEMM input of a bf16 run), [B, 2B)
    // embedding rows copied from the table (exact in the table's dtype: their remainder plane stays zero)
    g.cp_in_t.ensure(cp_rows * H * es * (split ? 2 : 1));
    const long lo_cp = split ? (long)(cp_rows * H) : 0;
    if (split) QTTS_HIP(hipMemsetAsync((char*)g.cp_in_t.p + (size_t)lo_cp * es, 0, cp_rows * H * es, st));
    g.cp_in_f.ensure(cp_rows * H * 4);
    g.trailing.ensure((size_t)B * Tt_max * H * 4);
    g.tts_pad.ensure((size_t)H * 4);
    g.toks_dec.ensure((size_t)B * sizeof(Tok));
    g.toks_cp.ensure((size_t)(2 * B + (size_t)B * G) * sizeof(Tok));
    g.row_len.ensure((size_t)B * 4);
    g.step.ensure((size_t)B * 4);
    g.eos_step.ensure((size_t)B * 4);
    g.active.ensure((size_t)B * 4);
    g.row_max.ensure((size_t)B * 4);
    g.hist.ensure((size_t)B * V * 4);
    g.uniforms.ensure((size_t)B * max_new * G * 4);
    g.codes.ensure((size_t)B * max_new * G * 4);
    g.trailing_len.ensure((size_t)B * 4);
    m.ensure_rope(m.talker, max_ctx + 1);
    m.ensure_rope(m.cp, G + 2);

    // KV pools
    {
        const int max_pages = (max_ctx + PAGE_TOKENS - 1) / PAGE_TOKENS;
        // every row owns max_pages consecutive pages: no page table, no dependent load in front of the K/V accesses
        const size_t page_elems = (size_t)2 * c.talker.nkv * PAGE_TOKENS * c.talker.d;
        const size_t npages = (size_t)B * max_pages;
        g.talker.pool.ensure((size_t)c.talker.layers * npages * page_elems * es);
        g.talker.kv = KVView{g.talker.pool.p, npages * page_elems, PAGE_TOKENS, c.talker.nkv, c.talker.d, max_pages, nullptr};
        QTTS_CHECK(G + 1 <= CP_PAGE_TOKENS, ERR_SHAPE, "generate: too many code groups for the code-predictor KV page");
        const size_t cpage = (size_t)2 * c.cp.nkv * CP_PAGE_TOKENS * c.cp.d;
        g.cp.pool.ensure((size_t)c.cp.layers * B * cpage * es);
        g.cp.kv = KVView{g.cp.pool.p, (size_t)B * cpage, CP_PAGE_TOKENS, c.cp.nkv, c.cp.d, 1, nullptr};
    }

    ScopedEvent ev0, ev1, ev2;
    QTTS_HIP(hipEventRecord(ev0, st));
    QTTS_HIP(hipMemsetAsync(g.active.p, 0, (size_t)B * 4, st));
    QTTS_HIP(hipMemsetAsync(g.step.p, 0, (size_t)B * 4, st));
    QTTS_HIP(hipMemsetAsync(g.row_len.p, 0, (size_t)B * 4, st));
    QTTS_HIP(hipMemsetAsync(g.eos_step.p, 0xff, (size_t)B * 4, st));
    QTTS_HIP(hipMemsetAsync(g.trailing_len.p, 0, (size_t)B * 4, st));
    QTTS_HIP(hipMemsetAsync(g.codes.p, 0, (size_t)B * max_new * G * 4, st));
    QTTS_HIP(hipMemsetAsync(g.uniforms.p, 0, (size_t)B * max_new * G * 4, st));
    // code-predictor token descriptors: pass 0 = 2B tokens (position-major), unmasked => both attend 2 keys
    // (code_predictor.rs:455-460 + attention/mod.rs:413-417); pass j = B tokens at position j + 1.
    {
        std::vector<Tok> ct((size_t)2 * B + (size_t)B * G);
        for (int p = 0; p < 2; ++p)
            for (int b = 0; b < B; ++b) ct[(size_t)p * B + b] = Tok{b, p, 0, 2};
        for (int j = 1; j < G - 1; ++j)
            for (int b = 0; b < B; ++b) ct[(size_t)2 * B + (size_t)j * B + b] = Tok{b, j + 1, 0, j + 2};
        QTTS_HIP(hipMemcpyAsync(g.toks_cp.p, ct.data(), ct.size() * sizeof(Tok), hipMemcpyHostToDevice, st));
        QTTS_HIP(hipStreamSynchronize(st));
    }
    void* hid_t = g.cp_in_t.p;                 // rows [0,B): post-norm hidden of each row's newest position
    float* hid_f = g.cp_in_f.as<float>();
    const float* tts_pad_row = g.tts_pad.as<float>();

    // ------------------------------------------------------------------ admission: requests ids[i] into rows slots[i]
    // (prompt embedding assembly, talker prefill into the row's KV pages, the row's sampling state).  Uses the activation
    // workspace, so it runs between frames; the rows that are decoding keep their state (hidden, KV, histories) untouched.
    // QTTS_CANONICAL_PREFILL=0: admission GEMMs through the kernels planned for their token count (faster for one or two short
    // prompts, but a prompt's state then depends on what was admitted with it, in the last bits)
    static const bool canonical = !(getenv("QTTS_CANONICAL_PREFILL") && atoi(getenv("QTTS_CANONICAL_PREFILL")) == 0);
    auto admit = [&](const std::vector<int>& ids, const std::vector<int>& rws) {
        struct Canon {
            Runner& r;
            Canon(Runner& rr, bool on) : r(rr) { r.canonical_rows = on; }
            ~Canon() { r.canonical_rows = false; }
        } canon(R, canonical);
        const int nb = (int)ids.size();
        PromptPlan pl;
        plan_prompts(c, reqs, ids, rws, pl);
        const int Ttot = (int)pl.p_toks.size(), Nt = (int)pl.text_ids.size();
        g.text_rows.ensure((size_t)Nt * H * 4);
        g.extra.ensure(std::max<size_t>(pl.extra_rows.size(), 1) * 4);
        g.ibuf.ensure(((size_t)Ttot * 3 + Nt + (size_t)nb * Tt_max + 4 * nb + 64) * 4);
        g.toks_prompt.ensure((size_t)Ttot * sizeof(Tok));
        // uniforms in the reference's draw order per utterance: frame by frame, cb0 first then codebooks 1..G-1
        // (generation.rs:1055-1060, code_predictor.rs:468-472,517-522); a greedy stage consumes no draw.
        if (shared_rng && (sp->do_sample || sp->subtalker_do_sample)) {
            // one stream for the whole batch in the reference's order (mt_rng.rs:22): per frame the talker draws of rows
            // 0 .. n-1, then codebook by codebook the draws of rows 0 .. n-1
            const qtts_request& r0 = reqs[ids[0]];
            uint64_t seed = r0.seed;
            if (!r0.use_seed)
                seed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
            HostMT rng((uint32_t)seed);
            std::vector<float> u((size_t)nb * max_new * G, 0.f);
            for (int s2 = 0; s2 < max_new; ++s2)
                for (int j = 0; j < G; ++j) {
                    if (!(j == 0 ? sp->do_sample : sp->subtalker_do_sample)) continue;
                    for (int i = 0; i < nb; ++i) u[((size_t)rws[(size_t)i] * max_new + s2) * G + j] = rng.uniform();
                }
            copy_sync(g.uniforms.p, u.data(), u.size() * 4, hipMemcpyHostToDevice, st);
        } else if (sp->do_sample || sp->subtalker_do_sample) {
            std::vector<float> u((size_t)max_new * G, 0.f);
            for (int i = 0; i < nb; ++i) {
                const qtts_request& rq = reqs[ids[(size_t)i]];
                uint64_t seed = rq.seed;
                if (!rq.use_seed)
                    seed = (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(
                               std::chrono::system_clock::now().time_since_epoch()).count() + (uint64_t)ids[(size_t)i] * 0x9E3779B97F4A7C15ull;
                HostMT rng((uint32_t)seed);
                std::fill(u.begin(), u.end(), 0.f);
                for (int s2 = 0; s2 < max_new; ++s2) {
                    float* row = &u[(size_t)s2 * G];
                    if (sp->do_sample) row[0] = rng.uniform();
                    if (sp->subtalker_do_sample)
                        for (int j = 1; j < G; ++j) row[j] = rng.uniform();
                }
                copy_sync(g.uniforms.as<float>() + (size_t)rws[(size_t)i] * max_new * G, u.data(), u.size() * 4, hipMemcpyHostToDevice, st);
            }
        }
        int* ib = g.ibuf.as<int>();
        int* d_ptext = ib;
        int* d_pcodec = ib + Ttot;
        int* d_pextra = ib + 2 * Ttot;
        int* d_textids = ib + 3 * Ttot;
        int* d_trail = d_textids + Nt;
        int* d_last_src = d_trail + nb * Tt_max;
        int* d_last_dst = d_last_src + nb;
        QTTS_HIP(hipMemcpyAsync(d_ptext, pl.p_text.data(), (size_t)Ttot * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipMemcpyAsync(d_pcodec, pl.p_codec.data(), (size_t)Ttot * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipMemcpyAsync(d_pextra, pl.p_extra.data(), (size_t)Ttot * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipMemcpyAsync(d_textids, pl.text_ids.data(), (size_t)Nt * 4, hipMemcpyHostToDevice, st));
        std::vector<int> trail_flat((size_t)nb * Tt_max, 2);
        for (int i = 0; i < nb; ++i)
            for (size_t k = 0; k < pl.trail_idx[(size_t)i].size(); ++k) trail_flat[(size_t)i * Tt_max + k] = pl.trail_idx[(size_t)i][k];
        QTTS_HIP(hipMemcpyAsync(d_trail, trail_flat.data(), trail_flat.size() * 4, hipMemcpyHostToDevice, st));
        QTTS_HIP(hipMemcpyAsync(g.toks_prompt.p, pl.p_toks.data(), (size_t)Ttot * sizeof(Tok), hipMemcpyHostToDevice, st));
        // every prefill chunk's tokens cut into runs of one row (tiles of the matrix-core attention kernel; m0 counts from the chunk)
        std::vector<AttnTile> ptiles, ctiles;
        std::vector<size_t> ptile_off;
        for (int o = 0; o < Ttot; o += chunk_rows) {
            build_attn_tiles(pl.p_toks.data() + o, std::min(chunk_rows, Ttot - o), ctiles);
            ptile_off.push_back(ptiles.size());
            ptiles.insert(ptiles.end(), ctiles.begin(), ctiles.end());
        }
        ptile_off.push_back(ptiles.size());
        g.tiles_prompt.ensure(std::max<size_t>(ptiles.size(), 1) * sizeof(AttnTile));
        QTTS_HIP(hipMemcpyAsync(g.tiles_prompt.p, ptiles.data(), ptiles.size() * sizeof(AttnTile), hipMemcpyHostToDevice, st));
        if (!pl.extra_rows.empty()) QTTS_HIP(hipMemcpyAsync(g.extra.p, pl.extra_rows.data(), pl.extra_rows.size() * 4, hipMemcpyHostToDevice, st));
        if (!pl.icl_rows.empty()) {  // sum_ref_code_embeddings (generation.rs:242-273) into the extra rows
            g.icl.ensure((pl.icl_codes.size() + pl.icl_rows.size()) * 4);
            QTTS_HIP(hipMemcpyAsync(g.icl.p, pl.icl_codes.data(), pl.icl_codes.size() * 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemcpyAsync(g.icl.as<int>() + pl.icl_codes.size(), pl.icl_rows.data(), pl.icl_rows.size() * 4, hipMemcpyHostToDevice, st));
            launch_sum_code_rows(st, dt, m.codec_emb, (const void* const*)m.cp_emb_ptrs.p, g.icl.as<int>(), g.icl.as<int>() + pl.icl_codes.size(),
                                 (int)pl.icl_rows.size(), G, H, g.extra.as<float>());
        }
        // the rows' sampling state
        std::vector<int> row_limit((size_t)nb);
        for (int i = 0; i < nb; ++i) {
            const int r = rws[(size_t)i];
            const int one = 1, zero = 0, none = -1;
            QTTS_HIP(hipMemcpyAsync(g.row_len.as<int>() + r, &pl.row_P[(size_t)i], 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemcpyAsync(g.trailing_len.as<int>() + r, &pl.trail_len[(size_t)i], 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemcpyAsync(g.step.as<int>() + r, &zero, 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemcpyAsync(g.eos_step.as<int>() + r, &none, 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemcpyAsync(g.active.as<int>() + r, &one, 4, hipMemcpyHostToDevice, st));
            row_limit[(size_t)i] = row_limit_of(ids[(size_t)i]);
            QTTS_HIP(hipMemcpyAsync(g.row_max.as<int>() + r, &row_limit[(size_t)i], 4, hipMemcpyHostToDevice, st));
            QTTS_HIP(hipMemsetAsync(g.hist.as<int>() + (size_t)r * V, 0, (size_t)V * 4, st));
            QTTS_HIP(hipMemsetAsync(g.codes.as<unsigned>() + (size_t)r * max_new * G, 0, (size_t)max_new * G * 4, st));
        }
        QTTS_HIP(hipStreamSynchronize(st));  // the small host values above live on this stack frame
        // text projection (talker.rs:345-348, mlp.rs:149-160)
        for (int o = 0; o < Nt; o += chunk_rows) {
            const int M = std::min(chunk_rows, Nt - o);
            launch_gather_rows(st, m.text_emb, dt, d_textids + o, g.ws.xn.p, dt, M, Ht, Ht);
            int ks = R.ks_for(M, Ht, Ht);
            R.gemm(m.fc1, g.ws.xn.p, M, Ht, Ht, ks, Ht);
            launch_bias_act(st, g.ws.part.as<float>(), ks, R.kstride(M, Ht), m.fc1_b, 1, g.ws.act.p, dt, M, Ht, g.ws.lo_act);
            ks = R.ks_for(M, H, Ht);
            R.gemm(m.fc2, g.ws.act.p, M, H, Ht, ks, Ht, g.ws.lo_act);
            launch_bias_act(st, g.ws.part.as<float>(), ks, R.kstride(M, H), m.fc2_b, 0, g.text_rows.as<float>() + (size_t)o * H, F32, M, H);
        }
        for (int i = 0; i < nb; ++i)
            launch_gather_rows(st, g.text_rows.p, F32, d_trail + (size_t)i * Tt_max, g.trailing.as<float>() + (size_t)rws[(size_t)i] * Tt_max * H, F32,
                               Tt_max, H, H);
        QTTS_HIP(hipMemcpyAsync(g.tts_pad.p, g.text_rows.as<float>() + (size_t)2 * H, (size_t)H * 4, hipMemcpyDeviceToDevice, st));
        // prefill (generation.rs:884-903)
        std::vector<int> last_src, last_dst;
        int row_end = 0;
        std::vector<int> ends((size_t)nb);
        for (int i = 0; i < nb; ++i) { row_end += pl.row_P[(size_t)i]; ends[(size_t)i] = row_end - 1; }
        for (int o = 0; o < Ttot; o += chunk_rows) {
            const int M = std::min(chunk_rows, Ttot - o);
            launch_assemble(st, g.text_rows.as<float>(), d_ptext + o, m.codec_emb, dt, d_pcodec + o, g.extra.as<float>(),
                            d_pextra + o, g.ws.x.as<float>(), M, H);
            launch_resid_norm(st, dt, g.ws.x.as<float>(), nullptr, 0, 0, nullptr, nullptr, m.talker.layers[0].ln1, c.talker.eps, g.ws.x.as<float>(),
                              g.ws.xn.p, nullptr, M, H, g.ws.lo_xn);
            // final norm output reuses act (T) and q (f32) as scratch rows
            R.tiles = g.tiles_prompt.as<AttnTile>() + ptile_off[(size_t)(o / chunk_rows)];
            R.n_tiles = (int)(ptile_off[(size_t)(o / chunk_rows) + 1] - ptile_off[(size_t)(o / chunk_rows)]);
            R.stack(m.talker, M, g.toks_prompt.as<Tok>() + o, g.talker.kv, g.ws.act.p, g.ws.q.as<float>(), true, false, g.ws.lo_act);
            R.tiles = nullptr;  // the code predictor's two-token prefill keeps the per-token kernel
            R.n_tiles = 0;
            last_src.clear();
            last_dst.clear();
            for (int i = 0; i < nb; ++i)
                if (ends[(size_t)i] >= o && ends[(size_t)i] < o + M) { last_src.push_back(ends[(size_t)i] - o); last_dst.push_back(rws[(size_t)i]); }
            if (!last_src.empty()) {
                const int nl = (int)last_src.size();
                QTTS_HIP(hipMemcpyAsync(d_last_src, last_src.data(), (size_t)nl * 4, hipMemcpyHostToDevice, st));
                QTTS_HIP(hipMemcpyAsync(d_last_dst, last_dst.data(), (size_t)nl * 4, hipMemcpyHostToDevice, st));
                launch_copy_rows(st, g.ws.act.p, d_last_src, hid_t, d_last_dst, dt, nl, H);
                if (split)
                    launch_copy_rows(st, (char*)g.ws.act.p + (size_t)g.ws.lo_act * es, d_last_src, (char*)hid_t + (size_t)lo_cp * es, d_last_dst, dt, nl, H);
                launch_copy_rows(st, g.ws.q.p, d_last_src, hid_f, d_last_dst, F32, nl, H);
                QTTS_HIP(hipStreamSynchronize(st));  // last_src/last_dst are reused by the next chunk
            }
        }
    };

    // development aid: QTTS_DEBUG_SCRUB="<mask>:<byte>" fills workspace buffers with a byte after every admission (bit 0 part,
    // 1 attn, 2 xn, 3 act, 4 ssA|ssB, 5 q, 6 x rows beyond the batch) - no kernel may depend on what a prefill left behind
    auto scrub = [&]() {
        const char* e = getenv("QTTS_DEBUG_SCRUB");
        if (!e) return;
        int mask = 0, byte = 0;
        if (sscanf(e, "%d:%d", &mask, &byte) != 2) return;
        if (mask & 1) QTTS_HIP(hipMemsetAsync(g.ws.part.p, byte, g.ws.part.bytes, st));
        if (mask & 2) QTTS_HIP(hipMemsetAsync(g.ws.attn.p, byte, g.ws.attn.bytes, st));
        if (mask & 4) QTTS_HIP(hipMemsetAsync(g.ws.xn.p, byte, g.ws.xn.bytes, st));
        if (mask & 8) QTTS_HIP(hipMemsetAsync(g.ws.act.p, byte, g.ws.act.bytes, st));
        if ((mask & 16) && g.ws.ssA.p) {
            QTTS_HIP(hipMemsetAsync(g.ws.ssA.p, byte, g.ws.ssA.bytes, st));
            QTTS_HIP(hipMemsetAsync(g.ws.ssB.p, byte, g.ws.ssB.bytes, st));
        }
        if (mask & 32) QTTS_HIP(hipMemsetAsync(g.ws.q.p, byte, g.ws.q.bytes, st));
        if (mask & 64) QTTS_HIP(hipMemsetAsync(g.ws.x.as<float>() + (size_t)B * Wmax, byte, g.ws.x.bytes - (size_t)B * Wmax * 4, st));
    };
    // the first B utterances take the rows in order
    int next_req = 0;
    std::vector<int> row_req((size_t)B, -1);
    {
        std::vector<int> ids, rws;
        for (int b = 0; b < B; ++b) { ids.push_back(next_req); rws.push_back(b); row_req[(size_t)b] = next_req++; }
        admit(ids, rws);
        scrub();
    }
    QTTS_HIP(hipEventRecord(ev1, st));

    // ------------------------------------------------------------------ frame pieces
    if (trace_c) g.trace_cp.ensure((size_t)B * max_new * (G - 1) * Vc * 4);
    if (trace_l) g.trace_logits.ensure((size_t)B * max_new * V * 4);
    if (trace_h) g.trace_hidden.ensure((size_t)B * max_new * H * 4);
    const long row_stride = (long)max_new * G;
    int* d_step = g.step.as<int>();

    // ---- tables of the code predictor's pass starts (CpMemo): built once per model, with this call's workspace and the decode
    // kernels, 64 codes at a time.  QTTS_CP_MEMO=0 keeps the projection + q|k|v launches.
    static const bool memo_off = getenv("QTTS_CP_MEMO") && atoi(getenv("QTTS_CP_MEMO")) == 0;
    const int cp_nqkv = (c.cp.nh + 2 * c.cp.nkv) * c.cp.d;
    const bool use_memo = fused_cp && !memo_off && m.proj && G > 2 && Vc % 64 == 0 && Hc % 4 == 0;
    if (use_memo) {
        CpMemo& mm = *m.cp_memo;
        std::lock_guard<std::mutex> lock(mm.mu);
        if (!mm.ready) {
            const int tables = G - 2;
            mm.x.alloc((size_t)tables * Vc * Hc * 4);
            mm.qkv.alloc((size_t)tables * Vc * cp_nqkv * 4);
            const LayerW& w0 = m.cp.layers[0];
            R.split(false);
            for (int t = 0; t < tables; ++t) {
                for (int r0 = 0; r0 < Vc; r0 += 64) {
                    const void* rows = (const char*)m.cp_emb[(size_t)t] + (size_t)r0 * H * es;
                    const int nt = R.k_split_resid(m.proj, rows, 0, 64, Hc, H, m.proj_b, false, w0.ln1, g.ws.ssA.as<float>());
                    FusedEpi e{};
                    e.mode = FUSED_PLAIN; e.ss_in = g.ws.ssA.as<float>(); e.ss_ld = g.ws.ss_ld; e.ntiles_in = nt; e.H_in = Hc; e.eps = c.cp.eps;
                    e.bias = w0.qkv_bias; e.out_f32 = g.ws.part.as<float>(); e.ld_out = cp_nqkv;
                    R.wide_gemm(w0.qkv, g.ws.xn.p, Hc, 0, 64, cp_nqkv, Hc, e);
                    const size_t row = (size_t)t * Vc + r0;
                    QTTS_HIP(hipMemcpyAsync(mm.x.as<float>() + row * Hc, g.ws.x.p, (size_t)64 * Hc * 4, hipMemcpyDeviceToDevice, st));
                    QTTS_HIP(hipMemcpyAsync(mm.qkv.as<float>() + row * cp_nqkv, g.ws.part.p, (size_t)64 * cp_nqkv * 4, hipMemcpyDeviceToDevice, st));
                }
            }
            QTTS_HIP(hipStreamSynchronize(st));
            R.split(true);
            mm.tables = tables; mm.Vc = Vc; mm.Hc = Hc; mm.nqkv = cp_nqkv;
            mm.ready = true;
        }
        g.cp_qkv_in.ensure((size_t)(B + 64) * cp_nqkv * 4);
    }
    // ---- single-token passes with every GEMM input in MFMA fragment order (k_gemm_frag.hip): no kernel stages its X tile through
    // LDS, the blocks are small (256 threads, a few KB of LDS) and those of several decode streams share a CU.  QTTS_CP_FRAG=0:
    // the five-launch layers of k_gemm_fused.hip for these passes too.
    static const bool frag_off = getenv("QTTS_CP_FRAG") && atoi(getenv("QTTS_CP_FRAG")) == 0;
    bool use_frag = use_memo && !frag_off && !split_cp && B <= 128 && c.cp.window == 0 && Runner::frag_layer_covered(m.cp) &&
                    gemm_frag_covers(Vc, Hc, FUSED_PLAIN) && (c.cp.nh * c.cp.d) % 64 == 0 && g.ws.ss_ld >= (B <= 64 ? 64 : 128);
    for (int j = 1; j < G - 1 && use_frag; ++j) use_frag = m.cp_head[(size_t)j].packed;
    if (use_frag) {
        auto ensure0 = [&](DevBuf& b, size_t n) {  // rows past the live ones are read by the GEMMs: they must hold finite values
            if (b.bytes < n) {
                b.alloc(n);
                QTTS_HIP(hipMemsetAsync(b.p, 0, n, st));
            }
        };
        ensure0(g.ws.xnp, (size_t)128 * Hc * 2);
        ensure0(g.ws.attnp, (size_t)128 * c.cp.nh * c.cp.d * 2);
        ensure0(g.ws.actp, (size_t)128 * c.cp.inter * 2);
    }
    // ---- single-token passes as ONE launch each (k_cp_pass.hip) where the stack has the shape its work split is written for.
    // Opt-in (QTTS_CP_PASS=1): measured 144 us per pass against 142 us for the five-launch layers on one stream (4.64 ms per frame
    // either way), and 1326 against 1614 audio-s/s with two sessions in flight - its 256 resident workgroups hold half of every
    // CU's registers while they wait for each other, which the other session's kernels then wait for.  Its workgroups wait for each
    // other, so at most two such launches may be in flight on a GPU (two workgroups fit a CU): one handle + one session.
    static const bool pass_on = getenv("QTTS_CP_PASS") && atoi(getenv("QTTS_CP_PASS")) == 1;
    bool use_pass = use_memo && pass_on && !split_cp && B <= CP_PASS_ROWS && Hc == CP_PASS_H && c.cp.inter == CP_PASS_I && c.cp.nh == CP_PASS_NH &&
                    c.cp.nkv == CP_PASS_NKV && c.cp.d == CP_PASS_D && c.cp.layers <= CP_PASS_MAX_LAYERS && c.cp.window == 0 && G <= 17 &&
                    (Vc == 1024 || Vc == 2048 || Vc == 4096);
    CpPassArgs pass{};
    if (use_pass) {
        for (int l = 0; l < c.cp.layers && use_pass; ++l) {
            const LayerW& w = m.cp.layers[(size_t)l];
            use_pass = w.qkv.packed && w.o.packed && w.gate_up_il.packed && w.down.packed;
            pass.layer[l] = CpPassLayer{(const bf16_t*)w.qkv.p, (const bf16_t*)w.o.p, (const bf16_t*)w.gate_up_il.p, (const bf16_t*)w.down.p,
                                        w.qkv_bias, w.o_bias, w.q_norm, w.k_norm, w.ln1, w.ln2};
        }
        for (int j = 1; j < G - 1 && use_pass; ++j) use_pass = m.cp_head[(size_t)j].packed;
    }
    if (use_pass) {
        pass.layers = c.cp.layers; pass.final_norm = m.cp.final_norm; pass.V = Vc; pass.eps = c.cp.eps;
        pass.scale = (float)std::pow((double)c.cp.d, -0.5);
        pass.M = B; pass.x_in = g.ws.x.as<float>(); pass.qkv0 = g.cp_qkv_in.as<float>(); pass.kv = g.cp.kv;
        pass.logits = g.ws.part.as<float>(); pass.ld_logits = Vc; pass.spin_limit = 1u << 22;
        const size_t nb = cp_pass_scratch_bytes();
        if (g.cp_pass.bytes < nb) {
            g.cp_pass.alloc(nb);
            QTTS_HIP(hipMemsetAsync(g.cp_pass.p, 0, nb, st));
        }
        cp_pass_bind_scratch(pass, g.cp_pass.p);
        static const bool want_ts = getenv("QTTS_CP_PASS_TS") != nullptr;
        if (want_ts) {
            g.cp_pass_ts.ensure(256 * 8);
            QTTS_HIP(hipMemsetAsync(g.cp_pass_ts.p, 0, 256 * 8, st));
            pass.ts = g.cp_pass_ts.as<unsigned long long>();
        }
    }
    // a wait between the workgroups of such a launch that timed out leaves its results and flags unusable
    auto check_pass = [&]() {
        if (!use_pass) return;
        unsigned err = 0;
        copy_sync(&err, pass.err, 4, hipMemcpyDeviceToHost, st);
        if (err) {
            QTTS_HIP(hipMemsetAsync(g.cp_pass.p, 0, CP_PASS_FLAG_BYTES, st));
            QTTS_HIP(hipStreamSynchronize(st));
            throw Error(ERR_HIP, "generate: the workgroups of a code-predictor pass stopped waiting for each other (the launch did not get the whole device)");
        }
    };

    // memo_table >= 0: the sampler also gathers the tabulated start of the next code-predictor pass for the code it drew
    auto sample = [&](const float* logits, int vocab, int gcol, bool talker, const void* next_table = nullptr, void* next_dst = nullptr,
                      int next_dst_dtype = F32, int memo_table = -1) {
        SampleArgs a{};
        if (memo_table >= 0) {
            const CpMemo& mm = *m.cp_memo;
            a.memo_x = mm.x.as<float>() + (size_t)memo_table * Vc * Hc; a.memo_x_dst = g.ws.x.as<float>(); a.memo_Hx = Hc;
            a.memo_q = mm.qkv.as<float>() + (size_t)memo_table * Vc * cp_nqkv; a.memo_q_dst = g.cp_qkv_in.as<float>(); a.memo_Nq = cp_nqkv;
        }
        a.next_table = next_table; a.next_dst = next_dst; a.next_H = H; a.next_table_bf16 = dt == BF16; a.next_dst_bf16 = next_dst_dtype == BF16;
        a.logits = logits; a.V = vocab; a.B = B;
        const double T = talker ? sp->temperature : sp->subtalker_temperature;
        a.temperature = T == 1.0 ? 1.0f : (float)(1.0 / T);
        a.top_k = talker ? sp->top_k : sp->subtalker_top_k;
        a.top_p = (float)(talker ? sp->top_p : sp->subtalker_top_p);
        a.do_sample = talker ? sp->do_sample : sp->subtalker_do_sample;
        a.rep_penalty = talker ? (float)sp->repetition_penalty : 1.0f;
        a.hist_count = talker && sp->repetition_penalty != 1.0 ? g.hist.as<int>() : nullptr;
        a.suppress_lo = talker ? std::max(0, V - 1024) : 0;
        a.suppress_hi = talker ? V : 0;
        a.eos_id = talker ? c.eos : -1;
        // sampling.rs:28-29 via generation.rs:841-842; subtalker: sampling.rs:117
        a.min_new_tokens = talker ? (dbg && dbg->min_new_tokens > 0 ? dbg->min_new_tokens : (sp->min_new_tokens < 0 ? 2 : sp->min_new_tokens)) : 0;
        a.step = d_step; a.active = g.active.as<int>(); a.max_steps = max_new; a.uniforms = g.uniforms.as<float>(); a.codes_base = g.codes.as<unsigned>(); a.row_stride = row_stride;
        a.G = G; a.g = gcol; a.eos_step = talker ? g.eos_step.as<int>() : nullptr;
        launch_sample(st, a);
    };

    // logits -> cb0 -> code predictor -> next talker input (generation.rs:1029-1106)
    // host_step: the frame index when the host knows it (eager launches; used by the code-predictor logit trace), -1 under capture
    auto frame_tail = [&](int host_step) {
        auto trace_cp = [&](int j) {
            if (!trace_c || host_step < 0) return;
            float* dst = g.trace_cp.as<float>() + ((size_t)host_step * (G - 1) + (size_t)j) * Vc;
            QTTS_HIP(hipMemcpy2DAsync(dst, (size_t)max_new * (G - 1) * Vc * 4, g.ws.part.p, (size_t)Vc * 4, (size_t)Vc * 4, B, hipMemcpyDeviceToDevice, st));
        };
        R.gemm(m.codec_head, hid_t, B, V, H, 1, H, lo_cp);
        // every sampler also copies the embedding row of the code it drew to where the next pass reads its input
        // (model dtype in front of the H -> Hc projection, f32 when the code predictor is as wide as the talker)
        void* in0 = m.proj ? (void*)((char*)g.cp_in_t.p + (size_t)B * H * es) : (void*)(g.cp_in_f.as<float>() + (size_t)B * H);
        void* inj = m.proj ? g.ws.act.p : g.ws.q.p;
        const int in_dt = m.proj ? dt : F32;
        sample(g.ws.part.as<float>(), V, 0, true, m.codec_emb, in0, in_dt);
        // ---- code predictor pass 0: [hidden, embed(cb0)] at positions 0, 1
        // Five layers deep, the code predictor stays within the bf16 tolerance with single-plane activations (split level 2 splits it too)
        R.split(split_cp);
        const Tok* ct = g.toks_cp.as<Tok>();
        if (fused_cp) {
            // five launches per layer: every GEMM applies its consumer's element-wise work (k_gemm_fused.hip)
            float* ssA = g.ws.ssA.as<float>();
            auto enter = [&](const void* rows_t, long rows_lo, const float* rows_f, int M) {  // pass input -> x, xn = T(x * ln1), ssA
                if (m.proj) return R.k_split_resid(m.proj, rows_t, rows_lo, M, Hc, H, m.proj_b, false, m.cp.layers[0].ln1, ssA);
                launch_rows_prep(st, dt, rows_f, m.cp.layers[0].ln1, g.ws.x.as<float>(), g.ws.xn.p, 0, ssA, M, Hc);
                return 1;
            };
            auto head = [&](int j, size_t row_off, int nt) {  // lm_head j on the newest position of every row (code_predictor.rs:464,514)
                FusedEpi e{};
                e.mode = FUSED_PLAIN; e.ss_in = ssA + row_off; e.ss_ld = g.ws.ss_ld; e.ntiles_in = nt; e.H_in = Hc; e.eps = c.cp.eps;
                e.out_f32 = g.ws.part.as<float>(); e.ld_out = Vc;
                R.timed(B, Vc, Hc, [&] { R.wide_gemm(m.cp_head[(size_t)j], (const char*)g.ws.xn.p + row_off * Hc * es, Hc, 0, B, Vc, Hc, e); });
                trace_cp(j);
            };
            int nt = enter(g.cp_in_t.p, lo_cp, g.cp_in_f.as<float>(), 2 * B);
            nt = R.stack_fused(m.cp, 2 * B, ct, g.cp.kv, nt, false);
            head(0, (size_t)B, nt);
            if (use_memo) sample(g.ws.part.as<float>(), Vc, 1, false, nullptr, nullptr, F32, 0);
            else sample(g.ws.part.as<float>(), Vc, 1, false, G > 2 ? m.cp_emb[0] : nullptr, inj, in_dt);
            for (int j = 1; j < G - 1; ++j) {  // code_predictor.rs:480-527
                // pass input: the embedding row of code j (exact in the model dtype, no remainder plane) -> x, xn, ssA; or, tabulated,
                // x and the first layer's q|k|v rows as the sampler gathered them
                if (use_frag && !use_pass) {
                    R.uni_len = j + 2;
                    for (int l = 0; l < c.cp.layers; ++l)
                        nt = R.layer_frag(m.cp, l, B, g.cp.kv, l, l + 1 == c.cp.layers ? m.cp.final_norm : m.cp.layers[(size_t)l + 1].ln1, nt,
                                          l == 0 ? g.cp_qkv_in.as<float>() : nullptr);
                    R.uni_len = 0;
                    FusedEpi e{};
                    e.mode = FUSED_PLAIN; e.ss_in = g.ws.ssA.as<float>(); e.ss_ld = g.ws.ss_ld; e.ntiles_in = nt; e.H_in = Hc; e.eps = c.cp.eps;
                    e.out_f32 = g.ws.part.as<float>(); e.ld_out = Vc;
                    R.timed(B, Vc, Hc, [&] { QTTS_CHECK(launch_gemm_frag(st, m.cp_head[(size_t)j], g.ws.xnp.p, Vc, Hc, e, B <= 64 ? 64 : 128), ERR_SHAPE, "generate: head shape"); });
                    trace_cp(j);
                } else if (use_pass) {  // layers + final norm + head of this pass in one launch
                    CpPassArgs pa = pass;
                    pa.pos = j + 1; pa.ctx = j + 2;
                    pa.cos = m.cp.cos.as<float>() + (size_t)(j + 1) * (c.cp.d / 2); pa.sin = m.cp.sin.as<float>() + (size_t)(j + 1) * (c.cp.d / 2);
                    pa.head = (const bf16_t*)m.cp_head[(size_t)j].p;
                    R.timed(B, Vc, Hc, [&] { launch_cp_pass(st, pa); });
                    trace_cp(j);
                } else {
                if (!use_memo) nt = enter(g.ws.act.p, 0, g.ws.q.as<float>(), B);
                R.uni_len = j + 2;  // every row: position j + 1, keys 0 .. j + 1 (what ct holds for this pass)
                nt = R.stack_fused(m.cp, B, ct + 2 * B + (size_t)j * B, g.cp.kv, nt, true, use_memo ? g.cp_qkv_in.as<float>() : nullptr);
                R.uni_len = 0;
                head(j, 0, nt);
                }
                if (use_memo) sample(g.ws.part.as<float>(), Vc, j + 1, false, nullptr, nullptr, F32, j + 1 < G - 1 ? j : -1);
                else sample(g.ws.part.as<float>(), Vc, j + 1, false, j + 1 < G - 1 ? m.cp_emb[(size_t)j] : nullptr, inj, in_dt);
            }
        } else {
        const long cxn = R.lo_xn;
        if (m.proj) {
            const int ks = R.ks_for(2 * B, Hc, H);
            R.gemm(m.proj, g.cp_in_t.p, 2 * B, Hc, H, ks, H, lo_cp);
            launch_resid_norm(st, dt, nullptr, g.ws.part.as<float>(), ks, R.kstride(2 * B, Hc), m.proj_b, nullptr, m.cp.layers[0].ln1, c.cp.eps,
                              g.ws.x.as<float>(), g.ws.xn.p, nullptr, 2 * B, Hc, cxn);
        } else {
            launch_resid_norm(st, dt, nullptr, g.cp_in_f.as<float>(), 1, 0, nullptr, nullptr, m.cp.layers[0].ln1, c.cp.eps, g.ws.x.as<float>(),
                              g.ws.xn.p, nullptr, 2 * B, Hc, cxn);
        }
        R.stack(m.cp, 2 * B, ct, g.cp.kv, g.ws.xn.p, nullptr, true, false, cxn);
        R.gemm(m.cp_head[0], (char*)g.ws.xn.p + (size_t)B * Hc * es, B, Vc, Hc, 1, Hc, cxn);
        trace_cp(0);
        sample(g.ws.part.as<float>(), Vc, 1, false, G > 2 ? m.cp_emb[0] : nullptr, inj, in_dt);
        for (int j = 1; j < G - 1; ++j) {  // code_predictor.rs:480-527
            if (m.proj) {
                const int ks = R.ks_for(B, Hc, H);
                R.gemm(m.proj, g.ws.act.p, B, Hc, H, ks, H);   // embedding rows: exact in the model dtype, no remainder plane
                launch_resid_norm(st, dt, nullptr, g.ws.part.as<float>(), ks, R.kstride(B, Hc), m.proj_b, nullptr, m.cp.layers[0].ln1, c.cp.eps,
                                  g.ws.x.as<float>(), g.ws.xn.p, nullptr, B, Hc, cxn);
            } else {
                launch_resid_norm(st, dt, nullptr, g.ws.q.as<float>(), 1, 0, nullptr, nullptr, m.cp.layers[0].ln1, c.cp.eps, g.ws.x.as<float>(), g.ws.xn.p,
                                  nullptr, B, Hc, cxn);
            }
            R.uni_len = j + 2;
            R.stack(m.cp, B, ct + 2 * B + (size_t)j * B, g.cp.kv, g.ws.xn.p, nullptr, true, true, cxn);
            R.uni_len = 0;
            R.gemm(m.cp_head[(size_t)j], g.ws.xn.p, B, Vc, Hc, 1, Hc, cxn);
            trace_cp(j);
            sample(g.ws.part.as<float>(), Vc, j + 1, false, j + 1 < G - 1 ? m.cp_emb[(size_t)j] : nullptr, inj, in_dt);
        }
        }
        // ---- next talker input (talker.rs:461-479 + generation.rs:1096-1106) and its layer-0 norm
        R.split(true);
        EmbedSumArgs ea{};
        ea.codec_emb = m.codec_emb; ea.cp_emb = (const void* const*)m.cp_emb_ptrs.p; ea.edtype = dt;
        ea.codes_base = g.codes.as<unsigned>(); ea.row_stride = row_stride; ea.step = d_step; ea.max_steps = max_new;
        ea.trailing = g.trailing.as<float>(); ea.trailing_len = g.trailing_len.as<int>(); ea.Tt_max = Tt_max;
        ea.tts_pad = tts_pad_row; ea.G = G; ea.H = H; ea.B = B;
        launch_embed_sum_norm(st, dt, ea, m.talker.layers[0].ln1, c.talker.eps, g.ws.x.as<float>(), g.ws.xn.p, g.ws.lo_xn);
        launch_advance_step(st, d_step, g.active.as<int>(), g.eos_step.as<int>(), B, max_new, streaming ? 1 : 0, g.row_max.as<int>());
    };
    auto talker_step = [&]() {  // generation.rs:1002-1026
        launch_make_decode_toks(st, g.row_len.as<int>(), g.toks_dec.as<Tok>(), B, c.talker.window, g.active.as<int>());
        R.stack(m.talker, B, g.toks_dec.as<Tok>(), g.talker.kv, hid_t, hid_f, true, true, lo_cp);
    };
    auto trace_copy = [&](int step) {
        // logits of this frame are recomputed into the partial buffer by frame_tail; hidden is in hid_f
        if (trace_h)
            QTTS_HIP(hipMemcpy2DAsync(g.trace_hidden.as<float>() + (size_t)step * H, (size_t)max_new * H * 4, hid_f, (size_t)H * 4,
                                      (size_t)H * 4, B, hipMemcpyDeviceToDevice, st));
        if (trace_l) {
            R.gemm(m.codec_head, hid_t, B, V, H, 1, H, lo_cp);
            QTTS_HIP(hipMemcpy2DAsync(g.trace_logits.as<float>() + (size_t)step * V, (size_t)max_new * V * 4, g.ws.part.p, (size_t)V * 4,
                                      (size_t)V * 4, B, hipMemcpyDeviceToDevice, st));
        }
    };

    // ------------------------------------------------------------------ frame loop
    std::vector<int> eos((size_t)B, -1), act_h((size_t)B, 1), step_h((size_t)B, 0);
    int steps_done = 0;
    auto poll = [&]() {
        QTTS_HIP(hipMemcpyAsync(eos.data(), g.eos_step.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        if (streaming) {
            QTTS_HIP(hipMemcpyAsync(act_h.data(), g.active.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
            QTTS_HIP(hipMemcpyAsync(step_h.data(), g.step.p, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        }
        QTTS_HIP(hipStreamSynchronize(st));
        check_pass();
    };
    auto all_eos = [&]() {
        poll();
        for (int b = 0; b < B; ++b)
            if (eos[(size_t)b] < 0) return false;
        return true;
    };
    // one graph per call: [talker step, frame tail] for the plain batch (its first tail runs eagerly on the prefill's hidden
    // state), [frame tail, talker step] for continuous batching, where a row enters between two launches with the hidden state of
    // its prefill in place
    auto capture = [&](bool tail_first) {
        if (g.graph_exec) { (void)hipGraphExecDestroy(g.graph_exec); g.graph_exec = nullptr; }
        hipGraph_t graph = nullptr;
        QTTS_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        try {
            if (tail_first) { frame_tail(-1); talker_step(); }
            else { talker_step(); frame_tail(-1); }
        } catch (...) {
            hipGraph_t junk = nullptr;
            (void)hipStreamEndCapture(st, &junk);
            if (junk) (void)hipGraphDestroy(junk);
            throw;
        }
        QTTS_HIP(hipStreamEndCapture(st, &graph));
        QTTS_HIP(hipGraphInstantiate(&g.graph_exec, graph, nullptr, nullptr, 0));
        QTTS_HIP(hipGraphDestroy(graph));
    };
    double launch_host_ms = 0;  // host time inside hipGraphLaunch (the runtime enqueues the graph's nodes one by one)
    auto launch_graph = [&]() {
        const auto t0 = std::chrono::steady_clock::now();
        QTTS_HIP(hipGraphLaunch(g.graph_exec, st));
        launch_host_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };

    if (streaming) {
        // ---------------------------------------------------------------- continuous batching
        std::vector<uint32_t> hrow((size_t)max_new * G);
        const auto done_cb = dbg ? dbg->done_cb : nullptr;
        const bool graphed = use_graph && max_new > 1;
        if (graphed) capture(true);
        int live = B, frames = 0;
        while (live > 0) {
            for (int k = 0; k < EOS_POLL; ++k) {
                if (graphed) launch_graph();
                else { frame_tail(frames); talker_step(); }
                ++frames;
            }
            poll();
            std::vector<int> ids, rws;
            for (int b = 0; b < B; ++b) {
                const int u = row_req[(size_t)b];
                if (u < 0 || act_h[(size_t)b] != 0) continue;
                // the utterance in row b is complete: what a batch of one returns for it (generation.rs:1125-1193)
                const int e = eos[(size_t)b];
                int num_steps = std::min(step_h[(size_t)b], max_new);
                if (e == 0) num_steps = 0;  // ended on its very first token: no frame at all (generation.rs:939,1145-1150)
                outs[u].num_steps = num_steps;
                outs[u].effective_len = (e >= 0 && e < num_steps) ? e : num_steps;
                if ((outs[u].codes || done_cb) && num_steps > 0)
                    copy_sync(hrow.data(), g.codes.as<unsigned>() + (size_t)b * max_new * G, (size_t)num_steps * G * 4, hipMemcpyDeviceToHost, st);
                if (outs[u].codes && num_steps > 0) std::memcpy(outs[u].codes, hrow.data(), (size_t)num_steps * G * 4);
                // the stream is idle here (poll() waited for it), so the callback may run other work of this handle on it
                if (done_cb) done_cb(dbg->done_user, u, hrow.data(), num_steps, outs[u].effective_len);
                row_req[(size_t)b] = -1;
                --live;
                if (next_req < n) { ids.push_back(next_req); rws.push_back(b); row_req[(size_t)b] = next_req++; ++live; }
            }
            if (!ids.empty()) { admit(ids, rws); scrub(); }
        }
        steps_done = frames;
        QTTS_HIP(hipEventRecord(ev2, st));
        QTTS_HIP(hipStreamSynchronize(st));
        QTTS_HIP(hipGetLastError());
    } else {
        // ---------------------------------------------------------------- the reference's batch
        if (tracing) trace_copy(0);
        frame_tail(0);
        steps_done = 1;
        if (use_graph && !tracing && max_new > 1) capture(false);
        // first-audio hook: hand the first frames of every row to the caller as soon as they exist
        bool chunk_sent = !(dbg && dbg->chunk_cb && dbg->first_frames > 0);
        auto send_chunk = [&](int frames) {
            chunk_sent = true;
            if (frames <= 0) return;
            std::vector<uint32_t> hc((size_t)B * frames * G);
            QTTS_HIP(hipMemcpy2DAsync(hc.data(), (size_t)frames * G * 4, g.codes.p, (size_t)max_new * G * 4, (size_t)frames * G * 4, B,
                                      hipMemcpyDeviceToHost, st));
            QTTS_HIP(hipStreamSynchronize(st));
            dbg->chunk_cb(dbg->chunk_user, hc.data(), B, frames);
        };
        bool done = false;
        while (steps_done < max_new && !done) {
            if (!chunk_sent && steps_done >= dbg->first_frames) send_chunk(dbg->first_frames);
            if (steps_done % EOS_POLL == 0 && all_eos()) { done = true; break; }
            if (g.graph_exec && use_graph && !tracing) {
                launch_graph();
            } else {
                talker_step();
                if (tracing) trace_copy(steps_done);
                frame_tail(steps_done);
            }
            ++steps_done;
        }
        QTTS_HIP(hipEventRecord(ev2, st));
        if (!chunk_sent) send_chunk(std::min(steps_done, dbg->first_frames));
        all_eos();
        QTTS_HIP(hipGetLastError());

        // results (generation.rs:1125-1193).  The reference stops one step after every row has produced EOS; rows keep
        // generating until then.
        bool every = true;
        int last_eos = -1;
        for (int b = 0; b < B; ++b) {
            if (eos[(size_t)b] < 0 || eos[(size_t)b] >= steps_done) { every = false; eos[(size_t)b] = -1; }
            else last_eos = std::max(last_eos, eos[(size_t)b]);
        }
        // every row ended on its very first token: the reference returns no frame at all (generation.rs:939,1145-1150)
        const int num_steps = every ? (last_eos == 0 ? 0 : std::min(steps_done, last_eos + 1)) : steps_done;
        std::vector<uint32_t> hcodes((size_t)B * max_new * G);
        copy_sync(hcodes.data(), g.codes.p, hcodes.size() * 4, hipMemcpyDeviceToHost, st);
        for (int b = 0; b < B; ++b) {
            outs[b].num_steps = num_steps;
            outs[b].effective_len = (eos[(size_t)b] >= 0 && eos[(size_t)b] < num_steps) ? eos[(size_t)b] : num_steps;
            if (outs[b].codes) std::memcpy(outs[b].codes, &hcodes[(size_t)b * max_new * G], (size_t)num_steps * G * 4);
        }
        if (trace_l) copy_sync(dbg->logits, g.trace_logits.p, (size_t)B * max_new * V * 4, hipMemcpyDeviceToHost, st);
        if (trace_h) copy_sync(dbg->hidden, g.trace_hidden.p, (size_t)B * max_new * H * 4, hipMemcpyDeviceToHost, st);
        if (trace_c) copy_sync(dbg->cp_logits, g.trace_cp.p, (size_t)B * max_new * (G - 1) * Vc * 4, hipMemcpyDeviceToHost, st);
    }
    if (use_pass && pass.ts) {  // marks of the last pass of the last frame
        std::vector<unsigned long long> h(256);
        copy_sync(h.data(), pass.ts, 256 * 8, hipMemcpyDeviceToHost, st);
        for (int w = 0; w < 2; ++w) {
            fprintf(stderr, "cp_pass marks, workgroup %d (us since the first):", w ? 255 : 0);
            for (int i = 0; i < 128 && h[(size_t)w * 128 + i]; ++i) fprintf(stderr, " %.2f", (double)(h[(size_t)w * 128 + i] - h[0]) * 0.01);
            fprintf(stderr, "\n");
        }
    }
    float ms01 = 0, ms12 = 0;
    QTTS_HIP(hipEventElapsedTime(&ms01, ev0, ev1));
    QTTS_HIP(hipEventElapsedTime(&ms12, ev1, ev2));
    m.timing[0] = ms01;
    m.timing[1] = ms12;
    m.timing[2] = g.timer.total_ms();
    m.timing[3] = (double)(g.timer.used / 2);
    m.timing[4] = g.timer.bytes;
    m.timing[5] = (double)steps_done;
    m.timing[7] = launch_host_ms;
    m.gemm_profile = g.timer.by_shape();
}

void run_generate(Model& m, const qtts_request* reqs, int n, const qtts_sampling* sp, int max_new, qtts_codes_out* outs,
                  const qtts_gen_debug* dbg) {
    generate_impl(m, reqs, n, sp, max_new, outs, dbg, 0);
}

void run_generate_continuous(Model& m, const qtts_request* reqs, int n, const qtts_sampling* sp, int max_new, int slots,
                             qtts_codes_out* outs, const qtts_gen_debug* dbg) {
    QTTS_CHECK(slots >= 1, ERR_BAD_ARG, "generate: at least one row slot");
    generate_impl(m, reqs, n, sp, max_new, outs, dbg, slots);
}

}  // namespace qtts
