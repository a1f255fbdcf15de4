// driver.cpp -- the speculative-decoding loops above llama_decode(), mirroring
//   chain driver   R/examples/speculative-simple/speculative-eagle.cpp:150-362  (decode_init / decode_initial /
//                  common_speculative_gen_draft R/common/speculative.cpp:137-299 / llama_decode / accept)
//   accept rule    common_sampler_sample_and_accept_n, R/common/sampling.cpp:423-451 (greedy: accept while the target's
//                  argmax equals the drafted token, then one bonus token)
//   KV fix-up      llama_kv_cache_seq_rm(ctx, 0, n_past, -1)  (speculative-eagle.cpp:355)
// and the plain (non-speculative) loop used as the 1x reference for the ">= 2x" target.
//
// Differences from the reference, all deliberate (bugs not to reproduce):
//   * the draft is fed the TARGET's feature of every accepted token at the start of a round (EAGLE's re-ingest),
//     not its own stale feature (A.5) and never uninitialised memory (A.1);
//   * prompt features come from the target's prompt pass.
#include "model.h"
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#define EH_API extern "C" __attribute__((visibility("default")))
using namespace eh;

static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// greedy pick on the host = llama_sampler_greedy_apply (R/src/llama-sampling.cpp): the FIRST of equal maxima.  GGML_OP_ARGMAX (the device-side
// pick of the greedy chain / verification) follows ggml_vec_argmax_f32 instead -- the LAST of equal maxima; the two differ on exact fp32 ties only
static int argmax(const float * v, int n) { int b = 0; float m = v[0]; for (int i = 1; i < n; ++i) if (v[i] > m) { m = v[i]; b = i; } return b; }
static float top_prob(const float * v, int n, int best) { double s = 0; const float m = v[best]; for (int i = 0; i < n; ++i) s += std::exp((double)(v[i] - m)); return (float)(1.0 / s); }

enum { ST_N_PREDICT, ST_N_DRAFTED, ST_N_ACCEPT, ST_N_ITERS, ST_T_PROMPT_US, ST_T_DECODE_US, ST_T_DRAFT_US, ST_T_VERIFY_US, ST_N_DRAFT_CALLS, ST_N_TARGET_CALLS, ST_COUNT };

struct SpecState {
    Model * tgt; Model * dft;
    int n_past = 0;                 // tokens whose K/V are valid in the target cache
    int32_t id_last = -1;           // sampled, not yet decoded by the target
    std::vector<int32_t> re_tok;    // accepted tokens of the previous round the draft has to (re-)ingest ...
    std::vector<float>   re_feat;   // ... with the target's features [re_tok.size()][n_embd]
    int re_pos0 = 0;                // position of re_tok[0]
    Batch bt, bd;
    std::vector<int32_t> drafts;
    bool fused_chain = getenv("EH_STEPWISE_DRAFT") == nullptr;
    bool device_argmax = getenv("EH_HOST_ARGMAX") == nullptr;      // EH_HOST_ARGMAX=1: fetch the logits and take the arg-max on the host like the reference's sampler
    bool device_tokens = getenv("EH_HOST_TOKENS") == nullptr;      // EH_HOST_TOKENS=1: the drafted tokens come back to the host before the verification batch is built
    bool chain_pending = false;                                    // the chain is still running: its tokens reach the target on the device
    bool re_on_device = false;                                     // re_feat was left on the device (Model::last_norm of the target); re_tok is on the host as always
};

// prompt: target over all tokens (features for the draft), draft over tokens 1..n-1
static int spec_prompt(SpecState & s, const int32_t * prompt, int n) {
    const int E = s.tgt->cfg.n_embd;
    // The device-side hand-offs (chain ids -> the target's GET_ROWS, the target's features / arg-max -> the chain's first step) pass raw
    // pointers into the other model's compute buffer and rely on ONE stream ordering the two models' graphs.  Models on different
    // Backend instances (different streams) take the host hand-off instead: drafted tokens and features cross through the host, which
    // waits for the producing backend first.
    if (s.dft && s.dft->be != s.tgt->be) s.device_tokens = false;
    s.tgt->kv.clear(); if (s.dft) s.dft->kv.clear();
    Batch & b = s.bt; b.clear();
    for (int i = 0; i < n; ++i) b.add(prompt[i], i, 0, true);
    int rc = s.tgt->decode(b, true);
    if (rc) return rc;
    s.n_past = n;
    s.id_last = argmax(s.tgt->logits_ith(n - 1), s.tgt->cfg.n_vocab);
    if (s.dft) {
        if (n > 1) {
            Batch & d = s.bd; d.clear();
            for (int i = 1; i < n; ++i) d.add(prompt[i], i, 0, i == n - 1);
            d.hidd.assign(s.tgt->hidden.begin(), s.tgt->hidden.begin() + (size_t)(n - 1) * E);     // F_0 .. F_{n-2}
            rc = s.dft->decode(d, false);
            if (rc) return rc;
        }
        s.re_tok.assign(1, s.id_last);
        s.re_feat.assign(s.tgt->hidden_ith(n - 1), s.tgt->hidden_ith(n - 1) + E);
        s.re_pos0 = n;
    }
    return 0;
}

// draft phase of a round (runs where the EAGLE head lives: rank 0 under tensor parallelism)
static int spec_draft(SpecState & s, int n_draft, float p_min, double * st, bool defer = false) {
    Model & T = *s.tgt, & D = *s.dft;
    const int E = T.cfg.n_embd, V = T.cfg.n_vocab;
    const double t0 = now_us();
    // ---- draft: re-ingest accepted tokens with true features, then autoregress on the draft's own features
    s.drafts.clear();
    D.kv.seq_rm(0, s.re_pos0, -1);
    Batch & d = s.bd; d.clear();
    const int nre = (int) s.re_tok.size();
    for (int i = 0; i < nre; ++i) d.add(s.re_tok[i], s.re_pos0 + i, 0, i == nre - 1);
    const bool chain = s.fused_chain && p_min <= 0.0f && n_draft > 1 && D.cfg.tp_size == 1;
    if (s.re_on_device && chain && T.last_norm.data && T.last_argmax.data) { D.first_feat = T.last_norm; D.first_ids = T.last_argmax; d.hidd.clear(); }
    else {
        if (s.re_on_device) { s.re_feat.resize((size_t) nre * E); if (T.fetch_last_hidden(s.re_feat.data(), nre)) return -9; s.re_on_device = false; }
        d.hidd = s.re_feat;
    }
    // greedy without a confidence cut-off: the whole chain runs as ONE graph with the token / feature hand-off on the device
    // (Model::decode_chain); EH_STEPWISE_DRAFT=1 keeps the reference's one-decode-per-step loop
    if (chain) {
        std::vector<int32_t> ids;
        const int rc = D.decode_chain(d, n_draft, ids, defer);
        if (rc == 0) {
            st[ST_N_DRAFT_CALLS] += 1; st[ST_N_DRAFTED] += n_draft;
            if (defer) {
                // the chain is running: prepare the verification batch (shape known: id_last + n_draft tokens at consecutive positions) meanwhile
                Batch & b = s.bt; b.clear();
                for (int i = 0; i <= n_draft; ++i) b.add(0, s.n_past + i, 0, true);
                s.tgt->want_logits = !s.device_argmax;
                // greedy verification needs no token on the host either: the target gathers its embeddings from the chain's id tensor
                // on the device (Model::dev_ids), so its pass goes out right behind the chain and ONE wait ends both
                const bool dev = s.device_tokens && s.device_argmax && D.chain_ids && T.cfg.tp_size == 1 && T.tok_embd_device();
                if (dev) { T.dev_ids = D.chain_ids; T.dev_table = T.tok_embd_device(); }
                const int rp = s.tgt->decode_prepare(b);
                s.tgt->want_logits = true;
                (void) rp;                                     // a failed preparation is simply redone (and reported) by decode()
                if (dev) { s.chain_pending = true; s.drafts.assign((size_t) n_draft, 0); st[ST_T_DRAFT_US] += now_us() - t0; return n_draft; }
                const int rw = D.chain_wait(ids);
                if (rw) return -10 + rw;
            }
            s.drafts = ids;
            st[ST_T_DRAFT_US] += now_us() - t0;
            return (int) s.drafts.size();
        }
        if (rc < 0) return -10 + rc;                       // rc == 1 (no KV room for the whole chain): fall through to the stepwise loop
        if (s.re_on_device) { s.re_feat.resize((size_t) nre * E); if (T.fetch_last_hidden(s.re_feat.data(), nre)) return -9; s.re_on_device = false; d.hidd = s.re_feat; }
    }
    // greedy without a confidence cut-off only needs the arg-max token: it is computed on the device (GGML_OP_ARGMAX appended to
    // the graph) and 4 bytes come back instead of a 128 KB logits row; p_min > 0 needs the probabilities, hence the logits
    D.want_logits = p_min > 0.0f || !s.device_argmax;
    for (int j = 0; j < n_draft; ++j) {
        int rc = D.decode(d, true);
        st[ST_N_DRAFT_CALLS] += 1;
        if (rc) return -10 - rc;
        const int last = d.n_tokens() - 1;
        const int id = D.argmax_ith(last);
        if (p_min > 0.0f && top_prob(D.logits_ith(last), V, id) < p_min) break;
        s.drafts.push_back(id);
        st[ST_N_DRAFTED] += 1;
        if (j + 1 == n_draft) break;
        const float * g = D.hidden_ith(last);
        const int pos = d.pos[last] + 1;
        std::vector<float> feat(g, g + E);
        d.clear(); d.add(id, pos, 0, true); d.hidd = std::move(feat);
    }
    D.want_logits = true;
    st[ST_T_DRAFT_US] += now_us() - t0;
    return (int) s.drafts.size();
}
// verify + accept + bookkeeping of a round; returns number of new tokens appended to out (>= 1) or < 0 on error
static int spec_verify(SpecState & s, int32_t * out, double * st) {
    Model & T = *s.tgt;
    const int E = T.cfg.n_embd;
    const double t1 = now_us();
    // ---- verify: [id_last, drafts...] in one target batch, logits for every token
    Batch & b = s.bt; b.clear();
    b.add(s.id_last, s.n_past, 0, true);
    for (size_t i = 0; i < s.drafts.size(); ++i) b.add(s.drafts[i], s.n_past + 1 + (int) i, 0, true);
    T.want_logits = !s.device_argmax;
    // the features of the accepted tokens go back to the draft on the device when the next round runs the fused chain (spec_draft)
    const bool keep_dev = s.chain_pending && s.device_tokens;
    int rc = T.decode(b, !keep_dev);
    T.want_logits = true; T.dev_ids = nullptr; T.dev_table = nullptr;
    st[ST_N_TARGET_CALLS] += 1;
    if (s.chain_pending) {                                         // the wait inside decode() ended the chain too: collect its tokens
        s.chain_pending = false;
        std::vector<int32_t> ids;
        const int rw = s.dft->chain_wait(ids);
        if (rw) return -10 + rw;
        s.drafts = ids;
    }
    if (rc) return -20 - rc;
    // ---- accept (greedy)
    int m = 0, n_out = 0;
    for (;;) {
        const int tok = T.argmax_ith(m);
        out[n_out++] = tok;
        if (m < (int) s.drafts.size() && tok == s.drafts[m]) { m++; continue; }
        break;
    }
    st[ST_N_ACCEPT] += m; st[ST_N_PREDICT] += n_out; st[ST_N_ITERS] += 1;
    // ---- bookkeeping: target keeps id_last + m drafts; next round re-ingests [d1..dm, bonus] with F rows 0..m
    s.re_tok.assign(out, out + n_out);
    s.re_on_device = keep_dev && T.last_norm.data && T.last_norm.rows >= n_out;
    if (!s.re_on_device) {
        s.re_feat.resize((size_t) n_out * E);
        if (keep_dev) { if (T.fetch_last_hidden(s.re_feat.data(), n_out)) return -29; }
        else for (int i = 0; i < n_out; ++i) memcpy(s.re_feat.data() + (size_t) i * E, T.hidden_ith(i), (size_t) E * 4);
    }
    s.re_pos0 = s.n_past + 1;
    s.n_past += m + 1;
    T.kv.seq_rm(0, s.n_past, -1);
    s.id_last = out[n_out - 1];
    st[ST_T_VERIFY_US] += now_us() - t1;
    return n_out;
}
static int spec_round(SpecState & s, int n_draft, float p_min, int32_t * out, double * st) {
    static const bool overlap = getenv("EH_NO_OVERLAP") == nullptr;      // prepare the verification graph while the draft chain runs
    const int nd = spec_draft(s, n_draft, p_min, st, overlap);
    if (nd < 0) return nd;
    return spec_verify(s, out, st);
}

EH_API void * eh_model_create(void * backend, const int * ci, float rms_eps, float rope_base, uint64_t seed, float accept_p, int predictable, void * target) {
    ModelConfig c;
    c.n_embd = ci[0]; c.n_head = ci[1]; c.n_head_kv = ci[2]; c.head_dim = ci[3]; c.n_ff = ci[4]; c.n_layer = ci[5]; c.n_vocab = ci[6];
    c.n_ctx = ci[7]; c.ftype = ci[8]; c.eagle = ci[9] != 0; c.rms_eps = rms_eps; c.rope_base = rope_base;
    c.tp_rank = ci[10]; c.tp_size = ci[11] > 0 ? ci[11] : 1;
    SynthOptions o; o.seed = seed; o.accept_p = accept_p; o.predictable = predictable != 0;
    return Model::create_synthetic((mh::Backend *) backend, c, o, (const Model *) target);
}
EH_API void eh_model_free(void * m) { delete (Model *) m; }
EH_API void eh_model_set_allreduce(void * m, Model::allreduce_fn fn, void * user) { ((Model *) m)->allreduce = fn; ((Model *) m)->allreduce_user = user; }
EH_API int64_t eh_model_n_allreduce(void * m) { return ((Model *) m)->n_allreduce; }
EH_API int64_t eh_model_weight_bytes(void * m) { return (int64_t) ((Model *) m)->weight_bytes; }
EH_API int eh_model_n_nodes(void * m) { return ((Model *) m)->last_n_nodes; }
// inspection of the LAST decode's graph (parity tooling: scripts/flip_replay.py, tests/test_teacher_forced_gpu.py).  The host's graph
// allocator never re-uses memory inside a graph, so every node a backend actually wrote is still there after the decode (run the plugin
// with GGML_MI355X_NO_FUSION=1 to have every node written).  Handles stay valid until the model's next decode.
EH_API void eh_model_force_layer_inputs(void * m, const float * const * rows, int n_layers) { Model * M = (Model *) m; M->forced_layer_inp.assign(rows, rows + n_layers); }
EH_API void * eh_model_node(void * m, int i) { Model * M = (Model *) m; return (M->gctx && i >= 0 && i < (int) M->gctx->nodes.size()) ? M->gctx->nodes[i] : nullptr; }
EH_API void eh_model_tensor_read(void * m, void * t, void * data, int64_t size) { ((Model *) m)->gctx->get((ggml_tensor *) t, data, 0, (size_t) size); }
EH_API int eh_model_decode(void * mp, int n, const int32_t * tok, const int32_t * pos, const int32_t * seq, const uint8_t * lg, const float * hidd, int want_hidden) {
    Model * m = (Model *) mp; Batch b;
    for (int i = 0; i < n; ++i) b.add(tok[i], pos[i], seq ? seq[i] : 0, lg ? lg[i] != 0 : true);
    if (hidd) b.hidd.assign(hidd, hidd + (size_t) n * m->cfg.n_embd);
    return m->decode(b, want_hidden != 0);
}
EH_API int eh_model_n_outputs(void * m) { return ((Model *) m)->n_outputs; }
EH_API const float * eh_model_logits(void * m) { return ((Model *) m)->logits.data(); }
EH_API const float * eh_model_hidden(void * m) { return ((Model *) m)->hidden.data(); }
EH_API void eh_model_kv_clear(void * m) { ((Model *) m)->kv.clear(); }
EH_API void eh_model_kv_seq_rm(void * m, int seq, int p0, int p1) { ((Model *) m)->kv.seq_rm(seq, p0, p1); }
EH_API void eh_model_kv_seq_cp(void * m, int a, int b, int p0, int p1) { ((Model *) m)->kv.seq_cp(a, b, p0, p1); }
EH_API void eh_model_kv_seq_keep(void * m, int seq) { ((Model *) m)->kv.seq_keep(seq); }
EH_API void eh_model_timers(void * mp, double * t) { Model * m = (Model *) mp; t[0] = m->t_build_us; t[1] = m->t_upload_us; t[2] = m->t_compute_us; t[3] = m->t_download_us; t[4] = (double) m->n_decode; t[5] = m->gctx->t_issue_us; t[6] = m->gctx->t_wait_us; }
EH_API void eh_model_timers_reset(void * mp) { Model * m = (Model *) mp; m->t_build_us = m->t_upload_us = m->t_compute_us = m->t_download_us = 0; m->n_decode = 0; m->gctx->t_issue_us = m->gctx->t_wait_us = 0; }

// Speculative generation.  stats: ST_* doubles.  Returns tokens generated, < 0 on error.
EH_API int eh_spec_run(void * tgt, void * dft, const int32_t * prompt, int n_prompt, int n_predict, int n_draft, float p_min,
                       int32_t * out_tokens, double * stats) {
    SpecState s; s.tgt = (Model *) tgt; s.dft = (Model *) dft;
    for (int i = 0; i < ST_COUNT; ++i) stats[i] = 0;
    const double t0 = now_us();
    int rc = spec_prompt(s, prompt, n_prompt);
    if (rc) return -1000 - rc;
    const double t1 = now_us();
    stats[ST_T_PROMPT_US] = t1 - t0;
    int n = 0;
    out_tokens[n++] = s.id_last; stats[ST_N_PREDICT] = 1;
    std::vector<int32_t> tmp((size_t) n_draft + 2);
    while (n < n_predict) {
        if (s.n_past + n_draft + 2 >= s.tgt->cfg.n_ctx) break;
        const int k = spec_round(s, n_draft, p_min, tmp.data(), stats);
        if (k < 0) return k;
        for (int i = 0; i < k && n < n_predict; ++i) out_tokens[n++] = tmp[i];
    }
    stats[ST_T_DECODE_US] = now_us() - t1;
    return n;
}
// one call = exactly `rounds` speculative rounds continuing from a primed state (bench.py times this)
struct SpecSession { SpecState s; std::vector<int32_t> tmp; };
EH_API void * eh_spec_begin(void * tgt, void * dft, const int32_t * prompt, int n_prompt) {
    SpecSession * ss = new SpecSession; ss->s.tgt = (Model *) tgt; ss->s.dft = (Model *) dft;
    if (spec_prompt(ss->s, prompt, n_prompt)) { delete ss; return nullptr; }
    return ss;
}
EH_API int eh_spec_rounds(void * sp, int rounds, int n_draft, float p_min, int32_t * out_tokens, int out_cap, double * stats) {
    SpecSession * ss = (SpecSession *) sp; ss->tmp.resize((size_t) n_draft + 2);
    for (int i = 0; i < ST_COUNT; ++i) stats[i] = 0;
    int n = 0;
    const double t0 = now_us();
    for (int r = 0; r < rounds; ++r) {
        if (ss->s.n_past + n_draft + 2 >= ss->s.tgt->cfg.n_ctx) return -5;
        const int k = spec_round(ss->s, n_draft, p_min, ss->tmp.data(), stats);
        if (k < 0) return k;
        for (int i = 0; i < k; ++i) if (n < out_cap) out_tokens[n++] = ss->tmp[i];
    }
    stats[ST_T_DECODE_US] = now_us() - t0;
    return n;
}
EH_API void eh_spec_end(void * sp) { delete (SpecSession *) sp; }
// the two halves of a round, for launchers that synchronise ranks in between (tensor parallel: bench_tp.py)
EH_API int eh_spec_draft(void * sp, int n_draft, float p_min, int32_t * drafts, double * stats) {
    SpecSession * ss = (SpecSession *) sp;
    const int n = spec_draft(ss->s, n_draft, p_min, stats);
    for (int i = 0; i < n; ++i) drafts[i] = ss->s.drafts[i];
    return n;
}
EH_API int eh_spec_verify(void * sp, int32_t * out_tokens, double * stats) { SpecSession * ss = (SpecSession *) sp; return spec_verify(ss->s, out_tokens, stats); }
EH_API void eh_spec_state(void * sp, int32_t * n_past, int32_t * id_last) { SpecSession * ss = (SpecSession *) sp; *n_past = ss->s.n_past; *id_last = ss->s.id_last; }

// Plain autoregressive decoding (what the speculative path must beat by >= 2x).  Like the speculative path it takes the greedy token with
// GGML_OP_ARGMAX on the device (4 bytes come back instead of a 128 KB logits row); EH_HOST_ARGMAX=1 fetches the logits and scans them on
// the host as the reference's sampler does.
EH_API int eh_plain_run(void * tgt, const int32_t * prompt, int n_prompt, int n_predict, int32_t * out_tokens, double * stats) {
    Model * T = (Model *) tgt;
    static const bool host_argmax = getenv("EH_HOST_ARGMAX") != nullptr;
    for (int i = 0; i < ST_COUNT; ++i) stats[i] = 0;
    const double t0 = now_us();
    T->kv.clear();
    Batch b;
    for (int i = 0; i < n_prompt; ++i) b.add(prompt[i], i, 0, i == n_prompt - 1);
    T->want_logits = host_argmax;
    int rc = T->decode(b, false);
    if (rc) { T->want_logits = true; return -1000 - rc; }
    const double t1 = now_us();
    stats[ST_T_PROMPT_US] = t1 - t0;
    int n = 0, n_past = n_prompt;
    int id = T->argmax_ith(n_prompt - 1);
    out_tokens[n++] = id;
    while (n < n_predict && n_past + 1 < T->cfg.n_ctx) {
        b.clear(); b.add(id, n_past, 0, true);
        rc = T->decode(b, false);
        stats[ST_N_TARGET_CALLS] += 1;
        if (rc) { T->want_logits = true; return -20 - rc; }
        n_past++;
        id = T->argmax_ith(0);
        out_tokens[n++] = id;
    }
    T->want_logits = true;
    stats[ST_N_PREDICT] = n; stats[ST_T_DECODE_US] = now_us() - t1;
    return n;
}
