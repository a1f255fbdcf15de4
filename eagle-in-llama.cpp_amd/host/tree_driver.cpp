// tree_driver.cpp -- the TREE speculative driver above llama_decode(), mirroring the control flow of
//   R/examples/speculative/speculative-eagle.cpp:232-670   (llama-speculative-eagle-tree)
//     verify / accept      :255-451   greedy: the target's token must equal the i-th token of an active branch (:396-421);
//                                     stochastic: r <= p_tgt / p_dft over randomly picked active branches, residual resampling (:261-394)
//     KV fix-up            :463-471   seq_keep(s_keep) -> seq_cp(s_keep -> 0) -> seq_keep(0) on both caches, seq_rm(tgt, s_keep, n_past_tgt, -1)
//     re-prime the draft   :474-495   the just-sampled token goes through the draft at n_past_dft
//     draft tree           :527-642   per depth, per drafting branch: candidate distribution; while cand[f].p > p_split and a branch id is
//                                     free: fork (seq_cp in the draft cache :561-562, the new id joins every target-batch token of the parent
//                                     :565-573); candidate k goes to branch k (:600-625); one draft decode per depth with one token per live
//                                     branch (:634); stop when the target batch holds more than n_draft tokens (:622, :639)
//     target evaluation    :646-656   seq_keep(tgt, 0); seq_cp(tgt, 0 -> s) for every branch id; ONE llama_decode over the whole tree
// The device sees none of this: a tree is only the seq-id sets of the KV cells, which llama_set_inputs turns into the additive mask
// (R/src/llama-context.cpp:136-210; mirrored in Model::decode).
//
// What differs from the reference, deliberately:
//   * the reference's tree driver never feeds the EAGLE head its hidden-state input (A.6: plain llama_decode on the draft context).  Here
//     a draft token is decoded together with the feature row that produced it: the target's result_norm row for the re-primed token,
//     the draft's own result_norm row of the parent for tree tokens (EAGLE-1 tree drafting) -- the same channel the chain driver uses.
//   * the candidate distribution is top-k -> softmax(logits / temp_dft): `temp_dft` may differ from the target's `temp`, so that a greedy
//     (temp = 0, deterministic) verification can be combined with a forking draft (at temp_dft = 0 the distribution is one-hot and the tree
//     degenerates to a chain, exactly as the reference does at --temp 0).  top_p / min_p / penalties of the reference's sampler chain are
//     not mirrored.
// Driver parity is UNPINNED: R/common cannot be built here (cmake-generated build-info.cpp), so no output of the reference driver exists
// to compare with; the tests compare this driver on the plugin with this driver on the reference CPU backend.
#include "model.h"
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <set>

#define EH_API extern "C" __attribute__((visibility("default")))
using namespace eh;

namespace {

inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Cand { int32_t id; float logit, p; };

// top-k by logit, then softmax over the survivors at temperature `temp` (temp <= 0: one-hot on the maximum, llama_sampler_temp_impl)
void candidates(const float * logits, int n_vocab, int top_k, float temp, std::vector<Cand> & out) {
    out.clear();
    const int k = std::max(1, std::min(top_k, n_vocab));
    std::vector<int32_t> idx((size_t) n_vocab);
    for (int i = 0; i < n_vocab; ++i) idx[i] = i;
    auto better = [&](int32_t a, int32_t b) { return logits[a] > logits[b] || (logits[a] == logits[b] && a < b); };
    std::partial_sort(idx.begin(), idx.begin() + k, idx.end(), better);
    out.resize((size_t) k);
    for (int i = 0; i < k; ++i) out[i] = { idx[i], logits[idx[i]], 0.0f };
    if (temp <= 0.0f) { out[0].p = 1.0f; return; }
    const float mx = out[0].logit;
    double sum = 0.0;
    for (auto & c : out) { c.p = std::exp((c.logit - mx) / temp); sum += c.p; }
    for (auto & c : out) c.p = (float)(c.p / sum);
}

// the same distribution from the k best (id, logit) pairs the device selected (descending, ties: lower id first -- the order of `better` above)
void candidates_from_topk(const int32_t * ids, const float * vals, int k, float temp, std::vector<Cand> & out) {
    out.clear();
    for (int i = 0; i < k && ids[i] >= 0; ++i) out.push_back({ ids[i], vals[i], 0.0f });
    if (out.empty()) return;
    if (temp <= 0.0f) { out[0].p = 1.0f; return; }
    const float mx = out[0].logit;
    double sum = 0.0;
    for (auto & c : out) { c.p = std::exp((c.logit - mx) / temp); sum += c.p; }
    for (auto & c : out) c.p = (float)(c.p / sum);
}

struct SeqDraft {                         // struct seq_draft, speculative-eagle.cpp:16-30
    bool active = false, drafting = false, skip = false;
    int i_batch_dft = 0;
    std::vector<int> i_batch_tgt;
    std::vector<int32_t> tokens;
    std::vector<std::vector<Cand>> dists;
    std::vector<float> feat;              // EAGLE: feature row that goes in with the branch's next draft token (result_norm of its parent)
};

enum { TS_N_PREDICT, TS_N_DRAFTED, TS_N_ACCEPT, TS_N_ITERS, TS_N_FORKS, TS_MAX_BATCH, TS_T_US, TS_T_DRAFT_US, TS_T_VERIFY_US, TS_N_DRAFT_CALLS, TS_COUNT };

struct TreeSession {
    Model * tgt = nullptr, * dft = nullptr;
    int n_seq_dft = 4, n_draft = 5, top_k = 40;
    float p_split = 0.1f, temp = 0.0f, temp_dft = 0.0f;
    std::mt19937 rng;
    std::uniform_real_distribution<float> u_dist{0.0f, 1.0f};
    int n_past_tgt = 0, n_past_dft = 0;
    std::vector<SeqDraft> drafts;
    Batch batch_tgt, batch_dft;
    bool primed = false;                  // a target batch has been evaluated and waits for verification
    bool dev_topk = true;                 // EH_HOST_TOPK=1: candidates from downloaded logits rows (the round-2 path)
    std::vector<Cand> dist_tgt;
};

// Prompt: target over all tokens (logits + features for every position), draft over tokens 1..n-1 with the features of their predecessors.
// Leaves the session as the reference does before its first loop iteration: drafts[0].i_batch_tgt = { last prompt row }.
int tree_prompt(TreeSession & S, const int32_t * prompt, int n) {
    Model & T = *S.tgt, & D = *S.dft;
    const int E = T.cfg.n_embd;
    T.kv.clear(); D.kv.clear();
    Batch & b = S.batch_tgt; b.clear();
    for (int i = 0; i < n; ++i) b.add(prompt[i], i, 0, true);
    T.want_logits = S.temp > 0.0f;
    int rc = T.decode(b, true);
    T.want_logits = true;
    if (rc) return rc;
    if (n > 1) {
        Batch & d = S.batch_dft; d.clear();
        for (int i = 1; i < n; ++i) d.add(prompt[i], i, 0, i == n - 1);
        d.hidd.assign(T.hidden.begin(), T.hidden.begin() + (size_t)(n - 1) * E);
        rc = D.decode(d, false);
        if (rc) return rc;
    }
    S.n_past_tgt = n; S.n_past_dft = n;
    S.drafts.assign((size_t) S.n_seq_dft, SeqDraft());
    S.drafts[0].active = true;
    S.drafts[0].i_batch_tgt.assign(1, n - 1);
    S.primed = true;
    return 0;
}

// One iteration of the reference's main loop: verify what the last target batch holds, fix the caches up, grow the next tree, evaluate it.
// Appends the tokens emitted by the verification (accepted drafts + the target's own token) to `out`.
int tree_round(TreeSession & S, std::vector<int32_t> & out, double * st) {
    Model & T = *S.tgt, & D = *S.dft;
    const int E = T.cfg.n_embd, V = T.cfg.n_vocab, n_seq_dft = S.n_seq_dft;
    auto & drafts = S.drafts;
    const double t0 = now_us();
    // ---------------------------------------------------------------- verify / accept (:255-451)
    std::set<int> active_seqs;
    for (int s = 0; s < n_seq_dft; ++s) if (drafts[s].active) active_seqs.insert(s);
    int i_dft = 0, s_keep = 0;
    int32_t token_id = -1;
    int row_of_token = -1;                 // target batch row whose logits produced token_id (its feature re-primes the draft)
    while (true) {
        bool accept = false;
        if (i_dft >= (int) drafts[s_keep].i_batch_tgt.size()) return -31;
        const int row = drafts[s_keep].i_batch_tgt[i_dft];
        const float * lg = S.temp > 0.0f ? T.logits_ith(row) : nullptr;
        if (S.temp > 0.0f && !lg) return -30;
        row_of_token = row;
        if (S.temp > 0.0f) {
            candidates(lg, V, S.top_k, S.temp, S.dist_tgt);
            auto & dist_tgt = S.dist_tgt;
            while (!active_seqs.empty()) {
                std::uniform_int_distribution<unsigned int> u_int(0, (unsigned) active_seqs.size() - 1);
                const int s = *std::next(active_seqs.begin(), u_int(S.rng));
                if (i_dft >= (int) drafts[s].tokens.size()) { drafts[s].active = false; active_seqs.erase(s); continue; }
                if (accept) {
                    if (drafts[s].tokens[i_dft] != drafts[s_keep].tokens[i_dft]) { drafts[s].active = false; active_seqs.erase(s); }
                    continue;
                }
                const float r = S.u_dist(S.rng);
                const std::vector<Cand> & dist_dft = drafts[s].dists[i_dft];
                float p_tgt = 0.0f, p_dft = 0.0f;
                for (const auto & c : dist_tgt) if (c.id == drafts[s].tokens[i_dft]) { p_tgt = c.p; break; }
                for (const auto & c : dist_dft) if (c.id == drafts[s].tokens[i_dft]) { p_dft = c.p; break; }
                if (r <= p_tgt / p_dft) {
                    s_keep = s; accept = true; token_id = drafts[s].tokens[i_dft];
                    break;
                }
                drafts[s].active = false;
                // residual distribution: p_tgt <- max(0, p_tgt - p_dft), renormalised (:343-371); matched by token id
                double sum = 0.0;
                for (auto & c : dist_tgt) { float pd = 0.0f; for (const auto & d : dist_dft) if (d.id == c.id) { pd = d.p; break; } c.p = std::max(0.0f, c.p - pd); sum += c.p; }
                if (sum > 0.0) for (auto & c : dist_tgt) c.p = (float)(c.p / sum);
                std::stable_sort(dist_tgt.begin(), dist_tgt.end(), [](const Cand & a, const Cand & b) { return a.p > b.p; });
                active_seqs.erase(s);
                for (int i = 0; i < n_seq_dft; ++i) {
                    if (i == s || (int) drafts[i].tokens.size() <= i_dft) continue;
                    if (drafts[i].tokens[i_dft] == drafts[s].tokens[i_dft]) { drafts[i].active = drafts[i].active && accept; if (!drafts[i].active) active_seqs.erase(i); }
                }
            }
            if (!accept) {                  // every draft rejected (or none left): sample the residual distribution
                std::vector<float> probs(dist_tgt.size());
                for (size_t i = 0; i < dist_tgt.size(); ++i) probs[i] = dist_tgt[i].p;
                std::discrete_distribution<> dd(probs.begin(), probs.end());
                token_id = dist_tgt[(size_t) dd(S.rng)].id;
            }
        } else {                            // greedy (:396-421)
            token_id = T.argmax_ith(row);          // GGML_OP_ARGMAX on the device, or the host scan when the logits were fetched
            if (token_id < 0) return -32;
            for (int s = 0; s < n_seq_dft; ++s) {
                if (!drafts[s].active) continue;
                if (i_dft < (int) drafts[s].tokens.size() && token_id == drafts[s].tokens[i_dft]) { s_keep = s; accept = true; }
                else drafts[s].active = false;
            }
        }
        out.push_back(token_id);
        st[TS_N_PREDICT] += 1;
        if (accept) { st[TS_N_ACCEPT] += 1; ++S.n_past_tgt; ++S.n_past_dft; ++i_dft; continue; }
        break;
    }
    st[TS_N_ITERS] += 1;
    // ---------------------------------------------------------------- KV fix-up (:463-471)
    D.kv.seq_keep(s_keep); D.kv.seq_cp(s_keep, 0, -1, -1); D.kv.seq_keep(0);
    T.kv.seq_rm(s_keep, S.n_past_tgt, -1); T.kv.seq_keep(s_keep); T.kv.seq_cp(s_keep, 0, -1, -1); T.kv.seq_keep(0);
    std::vector<float> feat_tok(T.hidden_ith(row_of_token), T.hidden_ith(row_of_token) + E);      // target feature behind token_id
    for (auto & d : drafts) { d.active = false; d.tokens.clear(); d.i_batch_tgt.clear(); d.dists.clear(); }
    drafts[0].tokens.push_back(token_id); drafts[0].dists.emplace_back(); drafts[0].i_batch_tgt.push_back(0);
    // ---------------------------------------------------------------- re-prime the draft with the sampled token (:488-495)
    const double t1 = now_us();
    D.want_logits = true;
    D.want_topk = S.dev_topk ? std::max(1, std::min(S.top_k, std::min(V, 64))) : 0;      // the draft's candidates: k best per row, selected on the device when the backend can
    S.batch_dft.clear(); S.batch_dft.add(token_id, S.n_past_dft, 0, true); S.batch_dft.hidd = feat_tok;
    D.kv.seq_rm(0, S.n_past_dft, -1);
    int rc = D.decode(S.batch_dft, true);
    st[TS_N_DRAFT_CALLS] += 1;
    if (rc) return -40 - rc;
    ++S.n_past_dft;
    // ---------------------------------------------------------------- grow the tree (:511-642)
    int n_seq_cur = 1, n_past_cur = S.n_past_dft;
    for (auto & d : drafts) { d.active = false; d.drafting = false; }
    drafts[0].active = true; drafts[0].drafting = true; drafts[0].i_batch_dft = 0;
    S.batch_tgt.clear();
    S.batch_tgt.add(drafts[0].tokens[0], S.n_past_tgt, 0, true);
    for (int i = 0; i < S.n_draft; ++i) {
        Batch next; std::vector<float> next_hidd;
        for (auto & d : drafts) d.skip = false;
        for (int s = 0; s < n_seq_dft; ++s) {
            if (!drafts[s].drafting || drafts[s].skip) continue;
            std::vector<Cand> cur_p;
            const float * hrow = D.hidden_ith(drafts[s].i_batch_dft);
            const int32_t * tk_ids; const float * tk_vals;
            if (!hrow) return -50;
            if (D.topk_ith(drafts[s].i_batch_dft, &tk_ids, &tk_vals)) candidates_from_topk(tk_ids, tk_vals, D.topk_k, S.temp_dft, cur_p);
            else {
                const float * lg = D.logits_ith(drafts[s].i_batch_dft);
                if (!lg) return -50;
                candidates(lg, V, S.top_k, S.temp_dft, cur_p);
            }
            { static const bool dbg = getenv("EH_TREE_DEBUG") != nullptr;
              if (dbg) { fprintf(stderr, "[tree] depth %d branch %d:", i, s); for (int f = 0; f < 4 && f < (int) cur_p.size(); ++f) fprintf(stderr, " id %d logit %.3f p %.4f |", cur_p[f].id, cur_p[f].logit, cur_p[f].p); fprintf(stderr, "\n"); } }
            std::vector<int> sa(1, s);
            for (int f = 1; f < 8 && f < (int) cur_p.size(); ++f) {
                if (n_seq_cur < n_seq_dft && cur_p[f].p > S.p_split) {
                    D.kv.seq_rm(n_seq_cur, -1, -1);
                    D.kv.seq_cp(s, n_seq_cur, -1, -1);
                    for (int t = 0; t < S.batch_tgt.n_tokens(); ++t) if (S.batch_tgt.seq_mask[t] & (1ull << s)) S.batch_tgt.seq_mask[t] |= 1ull << n_seq_cur;
                    SeqDraft & nd = drafts[n_seq_cur];
                    nd.active = true; nd.drafting = true; nd.skip = true;
                    nd.tokens = drafts[s].tokens; nd.dists = drafts[s].dists; nd.i_batch_dft = drafts[s].i_batch_dft; nd.i_batch_tgt = drafts[s].i_batch_tgt;
                    sa.push_back(n_seq_cur);
                    n_seq_cur++; st[TS_N_FORKS] += 1;
                } else break;
            }
            for (int is = 0; is < (int) sa.size(); ++is) {
                const int32_t id = cur_p[is].id;
                const int sq = sa[is];
                drafts[sq].tokens.push_back(id);
                drafts[sq].dists.push_back(cur_p);
                drafts[sq].i_batch_tgt.push_back(S.batch_tgt.n_tokens());
                S.batch_tgt.add(id, S.n_past_tgt + i + 1, sq, true);
                drafts[sq].i_batch_dft = next.n_tokens();
                next.add(id, n_past_cur, sq, true);
                next_hidd.insert(next_hidd.end(), hrow, hrow + E);           // the feature that produced this candidate distribution
                if (S.batch_tgt.n_tokens() > S.n_draft) drafts[sq].drafting = false;
            }
        }
        if (next.n_tokens() == 0) break;
        next.hidd = std::move(next_hidd);
        S.batch_dft = next;
        rc = D.decode(S.batch_dft, true);
        st[TS_N_DRAFT_CALLS] += 1;
        if (rc) return -60 - rc;
        ++n_past_cur; st[TS_N_DRAFTED] += S.batch_dft.n_tokens();
        if (S.batch_tgt.n_tokens() > S.n_draft) break;
    }
    D.want_topk = 0;
    const double t2 = now_us();
    st[TS_T_DRAFT_US] += t2 - t1;
    // ---------------------------------------------------------------- evaluate the tree on the target (:646-656)
    T.kv.seq_keep(0);
    for (int s = 1; s < n_seq_dft; ++s) T.kv.seq_cp(0, s, -1, -1);
    T.want_logits = S.temp > 0.0f;           // greedy verification only needs the arg-max row ids
    rc = T.decode(S.batch_tgt, true);
    T.want_logits = true;
    if (rc) return -70 - rc;
    ++S.n_past_tgt;
    st[TS_MAX_BATCH] = std::max(st[TS_MAX_BATCH], (double) S.batch_tgt.n_tokens());
    // the re-primed token leaves the branches' token lists (:659-670) but not i_batch_tgt: row 0 -- its logits -- is where the next
    // verification starts, then the row of the first drafted token, and so on
    for (int s = 0; s < n_seq_dft; ++s) {
        if (!drafts[s].active) continue;
        if (!drafts[s].tokens.empty()) { drafts[s].tokens.erase(drafts[s].tokens.begin()); drafts[s].dists.erase(drafts[s].dists.begin()); }
    }
    st[TS_T_VERIFY_US] += now_us() - t2;
    st[TS_T_US] += now_us() - t0;
    return 0;
}

} // namespace

// params: [n_seq_dft, n_draft, top_k, seed], fparams: [p_split, temp, temp_dft]
EH_API void * eh_tree_begin(void * tgt, void * dft, const int32_t * prompt, int n_prompt, const int32_t * ip, const float * fp) {
    TreeSession * S = new TreeSession;
    S->tgt = (Model *) tgt; S->dft = (Model *) dft;
    S->n_seq_dft = std::max(1, std::min(ip[0], S->tgt->cfg.n_seq_max)); S->n_draft = ip[1]; S->top_k = ip[2]; S->rng.seed((unsigned) ip[3]);
    S->p_split = fp[0]; S->temp = fp[1]; S->temp_dft = fp[2];
    S->dev_topk = getenv("EH_HOST_TOPK") == nullptr && S->top_k <= 64;
    if (tree_prompt(*S, prompt, n_prompt)) { delete S; return nullptr; }
    return S;
}
// runs rounds until `n_predict` tokens were emitted (or `max_rounds`); returns the number of tokens written to out (< 0: error)
EH_API int eh_tree_run(void * sp, int n_predict, int max_rounds, int32_t * out_tokens, int out_cap, double * stats) {
    TreeSession & S = *(TreeSession *) sp;
    for (int i = 0; i < TS_COUNT; ++i) stats[i] = 0;
    std::vector<int32_t> out;
    for (int r = 0; r < max_rounds && (int) out.size() < n_predict; ++r) {
        if (S.n_past_tgt + S.n_draft + S.n_seq_dft + 2 >= S.tgt->cfg.n_ctx) break;
        const int rc = tree_round(S, out, stats);
        if (rc) return rc;
    }
    const int n = std::min((int) out.size(), out_cap);
    memcpy(out_tokens, out.data(), (size_t) n * 4);
    return n;
}
EH_API void eh_tree_end(void * sp) { delete (TreeSession *) sp; }
