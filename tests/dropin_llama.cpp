// tests/dropin_llama.cpp -- drop-in proof under the REAL reference stack (test infrastructure only).
//
// Compiled in the build container against the reference's own headers (R/include/llama.h, R/ggml/include/*.h) and
// linked with oracle/_ref/libllama-ref.so + libggml-ref.so (the reference's src/*.cpp and ggml, built unmodified by
// oracle/Makefile); the binary lands in oracle/_ref/ and runs on the GPU box.
//
// It writes a small synthetic GGUF pair (llama target + eagle draft head, random weights quantised by the reference's
// own ggml_quantize_chunk), loads our plugin through the reference's loader (ggml_backend_load), and runs the
// reference's llama_decode() / llama_decode_initial() / llama_decode_draft() twice: all layers on the MI355X device
// (-ngl 99 equivalent) and all on the reference CPU backend.  The reference's scheduler, graph allocator (memory
// re-use!), KV cache, mask builder and hidden-state channel are all live; only graph_compute is ours.
// Output: one line per check with the relative error; exit code 0 iff every check passes.
//
// Weights: the residual branches (attn_output, ffn_down) are scaled so that a layer perturbs the residual stream by a
// few percent, as in a trained network.  With O(1) random branches the model is chaotic: int8 activation rounding
// turns a 1e-7 difference in fp32 summation order into ~sqrt(eps) output noise per quantised mat-mul (measured with
// DROPIN_TRACE=1: 2e-7 -> 5e-4 -> 2e-3 -> 7e-3 over one layer, for the reference's own AVX2-vs-scalar builds alike),
// which says nothing about the backend under test.
#include "llama.h"
#include "ggml.h"
#include "ggml-backend.h"
#include "gguf.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

struct dims { int n_embd, n_head, n_head_kv, n_ff, n_layer, n_vocab, n_ctx; };

static void add_tensor(gguf_context * g, ggml_context * ctx, std::mt19937 & rng, const char * name, ggml_type type, int64_t ne0, int64_t ne1, float scale, bool ones = false) {
    ggml_tensor * t = ne1 > 0 ? ggml_new_tensor_2d(ctx, type, ne0, ne1) : ggml_new_tensor_1d(ctx, type, ne0);
    ggml_set_name(t, name);
    const int64_t n = ggml_nelements(t);
    std::vector<float> f(n);
    std::normal_distribution<float> nd(0.0f, scale);
    for (auto & v : f) v = ones ? 1.0f : nd(rng);
    if (type == GGML_TYPE_F32) memcpy(t->data, f.data(), n * 4);
    else ggml_quantize_chunk(type, f.data(), t->data, 0, ne1 > 0 ? ne1 : 1, ne0, nullptr);
    gguf_add_tensor(g, t);
}

static void write_model(const char * path, const dims & d, bool eagle, ggml_type wtype, unsigned seed) {
    gguf_context * g = gguf_init_empty();
    const char * arch = eagle ? "eagle" : "llama";
    auto key = [&](const char * suffix) { return std::string(arch) + "." + suffix; };
    gguf_set_val_str(g, "general.architecture", arch);
    gguf_set_val_str(g, "general.name", eagle ? "synthetic-eagle" : "synthetic-llama");
    gguf_set_val_u32(g, key("context_length").c_str(), d.n_ctx);
    gguf_set_val_u32(g, key("embedding_length").c_str(), d.n_embd);
    gguf_set_val_u32(g, key("block_count").c_str(), eagle ? 1 : d.n_layer);
    gguf_set_val_u32(g, key("feed_forward_length").c_str(), d.n_ff);
    gguf_set_val_u32(g, key("attention.head_count").c_str(), d.n_head);
    gguf_set_val_u32(g, key("attention.head_count_kv").c_str(), d.n_head_kv);
    gguf_set_val_u32(g, key("rope.dimension_count").c_str(), d.n_embd / d.n_head);
    gguf_set_val_f32(g, key("attention.layer_norm_rms_epsilon").c_str(), 1e-6f);
    gguf_set_val_u32(g, key("vocab_size").c_str(), d.n_vocab);
    gguf_set_val_str(g, "tokenizer.ggml.model", "no_vocab");
    const size_t mem = (size_t) 64 << 20;
    ggml_init_params ip = { mem * 8, nullptr, false };
    ggml_context * ctx = ggml_init(ip);
    std::mt19937 rng(seed);
    const int hd = d.n_embd / d.n_head, kv = hd * d.n_head_kv;
    char nm[128];
    add_tensor(g, ctx, rng, "token_embd.weight", GGML_TYPE_F32, d.n_embd, d.n_vocab, 1.0f);
    if (!eagle) {
        add_tensor(g, ctx, rng, "output_norm.weight", GGML_TYPE_F32, d.n_embd, 0, 0, true);
        add_tensor(g, ctx, rng, "output.weight", GGML_TYPE_Q6_K, d.n_embd, d.n_vocab, 0.05f);
    } else {
        add_tensor(g, ctx, rng, "fc.weight", wtype, 2 * d.n_embd, d.n_embd, 0.03f);
        add_tensor(g, ctx, rng, "fc.bias", GGML_TYPE_F32, d.n_embd, 0, 0.1f);
    }
    for (int i = 0; i < (eagle ? 1 : d.n_layer); ++i) {
        if (!eagle) { snprintf(nm, sizeof nm, "blk.%d.attn_norm.weight", i); add_tensor(g, ctx, rng, nm, GGML_TYPE_F32, d.n_embd, 0, 0, true); }
        snprintf(nm, sizeof nm, "blk.%d.attn_q.weight", i);      add_tensor(g, ctx, rng, nm, wtype, d.n_embd, d.n_embd, 0.05f);
        snprintf(nm, sizeof nm, "blk.%d.attn_k.weight", i);      add_tensor(g, ctx, rng, nm, wtype, d.n_embd, kv, 0.05f);
        snprintf(nm, sizeof nm, "blk.%d.attn_v.weight", i);      add_tensor(g, ctx, rng, nm, (i % 2) ? GGML_TYPE_Q6_K : wtype, d.n_embd, kv, 0.05f);
        snprintf(nm, sizeof nm, "blk.%d.attn_output.weight", i); add_tensor(g, ctx, rng, nm, wtype, d.n_embd, d.n_embd, 0.004f);
        snprintf(nm, sizeof nm, "blk.%d.ffn_norm.weight", i);    add_tensor(g, ctx, rng, nm, GGML_TYPE_F32, d.n_embd, 0, 0, true);
        snprintf(nm, sizeof nm, "blk.%d.ffn_gate.weight", i);    add_tensor(g, ctx, rng, nm, wtype, d.n_embd, d.n_ff, 0.05f);
        snprintf(nm, sizeof nm, "blk.%d.ffn_up.weight", i);      add_tensor(g, ctx, rng, nm, wtype, d.n_embd, d.n_ff, 0.05f);
        snprintf(nm, sizeof nm, "blk.%d.ffn_down.weight", i);    add_tensor(g, ctx, rng, nm, (i % 2) ? GGML_TYPE_Q6_K : wtype, d.n_ff, d.n_embd, 0.004f);
    }
    gguf_write_to_file(g, path, false);
    gguf_free(g); ggml_free(ctx);
}

struct run_out { std::vector<float> prompt_logits, step_logits, tree_logits, long_logits, draft_logits; std::vector<std::pair<std::string, std::vector<float>>> trace; };

// DROPIN_TRACE=1: record every f32 node of the first target decode through the scheduler's eval callback.
// DROPIN_CUT=Qcur,ffn_gate: record only the nodes whose names start with one of the prefixes.  The reference scheduler then hands the
// backend graph VIEWS that end at those nodes (ggml-backend.cpp:1402-1430) and the host reads them: with "Qcur" a view ends between wq
// and wk, so a norm folded into wq's launch must be in memory for the next view's wk / wv -- the cut ADVICE r1 / VERDICT r2 #13 describe.
static bool g_trace = false;
static std::vector<std::string> g_cut;
static run_out * g_cur = nullptr;
static bool trace_cb(struct ggml_tensor * t, bool ask, void * user) {
    (void) user;
    if (!g_trace || !g_cur) return false;
    if (!g_cut.empty()) {
        bool hit = false;
        for (const auto & p : g_cut) if (!strncmp(t->name, p.c_str(), p.size()) && t->name[p.size()] == '-') hit = true;
        if (!hit || t->type != GGML_TYPE_F32) return false;
        if (ask) return true;
    } else
    if (ask) return t->type == GGML_TYPE_F32 && g_cur->trace.size() < 400;
    std::vector<float> v(ggml_nelements(t));
    if (ggml_is_contiguous(t)) ggml_backend_tensor_get(t, v.data(), 0, ggml_nbytes(t));
    g_cur->trace.emplace_back(std::string(t->name) + " [" + ggml_op_name(t->op) + "]", std::move(v));
    return true;
}

static std::vector<float> grab(llama_context * c, int i, int n_vocab) { const float * p = llama_get_logits_ith(c, i); return std::vector<float>(p, p + n_vocab); }

static bool run(const std::string & tgt_path, const std::string & dft_path, const dims & d, int ngl, run_out & o) {
    llama_model_params mp = llama_model_default_params();
    mp.n_gpu_layers = ngl; mp.use_mmap = true;
    // DROPIN_SPLIT_ROW=1: -sm row.  The reference's loader (llama-model.cpp:304-326) asks the plugin's registry for
    // "ggml_backend_split_buffer_type" and places the mat-mul weights in it; with GGML_MI355X_SPLIT_FAKE_DEVICES=3 the plugin slices
    // every weight over three logical devices of the one GPU (csrc/split.cpp)
    if (ngl > 0 && getenv("DROPIN_SPLIT_ROW")) mp.split_mode = LLAMA_SPLIT_MODE_ROW;
    llama_model * mt = llama_model_load_from_file(tgt_path.c_str(), mp);
    llama_model * md = llama_model_load_from_file(dft_path.c_str(), mp);
    if (!mt || !md) { fprintf(stderr, "model load failed\n"); return false; }
    llama_context_params cp = llama_context_default_params();
    cp.n_ctx = d.n_ctx; cp.n_batch = 64; cp.n_ubatch = 64; cp.n_seq_max = 4; cp.embeddings = true;    // the fork only yields logits with embeddings on
    cp.n_threads = 4; cp.n_threads_batch = 4;
    cp.flash_attn = getenv("DROPIN_FLASH_ATTN") != nullptr;     // -fa: GGML_OP_FLASH_ATTN_EXT, row-major V cache, f16 mask, n_kv padded to 256
    g_trace = getenv("DROPIN_TRACE") != nullptr || getenv("DROPIN_CUT") != nullptr; g_cur = &o;
    g_cut.clear();
    if (const char * c = getenv("DROPIN_CUT")) { std::string v(c); size_t a = 0; while (a <= v.size()) { const size_t b = v.find(',', a); const std::string tok = v.substr(a, b == std::string::npos ? std::string::npos : b - a); if (!tok.empty()) g_cut.push_back(tok); if (b == std::string::npos) break; a = b + 1; } }
    if (g_trace) { cp.cb_eval = trace_cb; cp.cb_eval_user_data = nullptr; }
    llama_context * ct = llama_init_from_model(mt, cp);
    llama_context * cd = llama_init_from_model(md, cp);
    if (!ct || !cd) { fprintf(stderr, "context init failed\n"); return false; }
    const int V = d.n_vocab;
    // 1. prompt: 12 tokens, logits for all
    llama_batch b = llama_batch_init(64, 0, 4);
    auto add = [&](int tok, int pos, std::vector<int> seqs, bool lg) { const int i = b.n_tokens; b.token[i] = tok; b.pos[i] = pos; b.n_seq_id[i] = (int) seqs.size(); for (size_t s = 0; s < seqs.size(); ++s) b.seq_id[i][s] = seqs[s]; b.logits[i] = lg; b.n_tokens++; };
    b.n_tokens = 0;
    for (int i = 0; i < 12; ++i) add(5 + i * 7 % V, i, {0}, true);
    if (llama_decode(ct, b) != 0) { fprintf(stderr, "llama_decode(prompt) failed\n"); return false; }
    for (int i = 0; i < 12; ++i) { auto v = grab(ct, i, V); o.prompt_logits.insert(o.prompt_logits.end(), v.begin(), v.end()); }
    if (g_cut.empty()) g_cur = nullptr;                         // DROPIN_TRACE: only the first decode; DROPIN_CUT: every decode of both contexts
    // 2. single-token step
    b.n_tokens = 0; add(33, 12, {0}, true);
    if (llama_decode(ct, b) != 0) return false;
    o.step_logits = grab(ct, 0, V);
    // 3. tree verify: two branches sharing the prefix (llama_kv_cache_seq_cp, as the tree driver does)
    llama_kv_cache_seq_cp(ct, 0, 1, -1, -1); llama_kv_cache_seq_cp(ct, 0, 2, -1, -1);
    b.n_tokens = 0; add(40, 13, {1}, true); add(41, 14, {1}, true); add(50, 13, {2}, true); add(51, 14, {2}, true); add(52, 15, {2}, true);
    if (llama_decode(ct, b) != 0) return false;
    for (int i = 0; i < 5; ++i) { auto v = grab(ct, i, V); o.tree_logits.insert(o.tree_logits.end(), v.begin(), v.end()); }
    // 3b. a 40-token batch on a fresh sequence: more than 24 tokens take the plugin's big-batch path (int8 GEMM over one shared activation
    //     image, norm and SwiGLU product folded into the quantiser launches) -- here under the reference's allocator, which re-uses the
    //     memory of gate / up / the norm input as soon as their last reader has run
    b.n_tokens = 0;
    for (int i = 0; i < 40; ++i) add(3 + (i * 11) % (V - 3), i, {3}, i >= 36);
    if (llama_decode(ct, b) != 0) { fprintf(stderr, "llama_decode(long batch) failed\n"); return false; }
    for (int i = 36; i < 40; ++i) { auto v = grab(ct, i, V); o.long_logits.insert(o.long_logits.end(), v.begin(), v.end()); }
    llama_kv_cache_seq_rm(ct, 3, -1, -1);
    // 4. EAGLE channel: target step that hands result_norm to the draft, then two draft steps on the draft's own feature.
    //    The draft context first ingests a 12-token batch with plain llama_decode (as the tree driver does for the prompt):
    //    besides priming its KV cache this sizes its output buffer, without which the reference's hidden-state pointer
    //    lands past the end of the allocation and corrupts the heap on ANY backend.
    // (DROPIN_CUT skips this part: the draft's prompt ingestion reads a hidden-state buffer the reference never initialises, SURVEY appendix
    //  A.1 -- stale heap memory, equal in two runs only as long as nothing else allocates, and the recorded tensors do)
    if (!g_cut.empty()) { llama_batch_free(b); llama_free(ct); llama_free(cd); llama_model_free(mt); llama_model_free(md); return true; }
    b.n_tokens = 0;
    for (int i = 0; i < 12; ++i) add(5 + i * 7 % V, i, {0}, true);
    if (llama_decode(cd, b) != 0) { fprintf(stderr, "llama_decode(draft prompt) failed\n"); return false; }
    b.n_tokens = 0; add(9, 12, {0}, true);                      // a 1-output call re-seats the hidden pointer INSIDE the (larger) buffer
    if (llama_decode(cd, b) != 0) return false;
    // The draft's K/V of that ingestion are dropped: plain llama_decode on an eagle graph uploads a hidden-state buffer nobody has filled
    // (SURVEY appendix A.1), so those cells hold whatever the allocation held before -- equal between two runs only by luck of the heap
    // (it stopped being equal the day the split buffer type's staging allocations moved things around).  The two draft steps below
    // then attend to their own cells only, which both backends compute from defined inputs.
    llama_kv_cache_clear(cd);
    llama_kv_cache_seq_keep(ct, 0);
    b.n_tokens = 0; add(60, 13, {0}, true);
    if (llama_decode_initial(ct, b, cd) != 0) { fprintf(stderr, "llama_decode_initial failed\n"); return false; }
    // the hand-off is an ASYNC read-back into the draft context's host buffer (llama.cpp:10424) that the draft's input upload then reads
    // with a synchronous tensor_set (llama-context.cpp:87): nothing orders the two.  The reference's example samples from the target's
    // logits in between (llama_get_logits_ith -> llama_synchronize); this program has to do the same or it races on ANY asynchronous backend
    llama_synchronize(ct);
    b.n_tokens = 0; add(61, 12, {0}, true);
    if (llama_decode_draft(cd, b, ct) != 0) { fprintf(stderr, "llama_decode_draft failed\n"); return false; }
    { auto v = grab(cd, 0, V); o.draft_logits.insert(o.draft_logits.end(), v.begin(), v.end()); }
    b.n_tokens = 0; add(62, 13, {0}, true);
    if (llama_decode_draft(cd, b, ct) != 0) return false;
    { auto v = grab(cd, 0, V); o.draft_logits.insert(o.draft_logits.end(), v.begin(), v.end()); }
    llama_batch_free(b);
    llama_free(ct); llama_free(cd); llama_model_free(mt); llama_model_free(md);
    return true;
}

static bool check(const char * what, const std::vector<float> & a, const std::vector<float> & b, int V, double tol) {
    if (a.size() != b.size() || a.empty()) { printf("%-14s size mismatch %zu vs %zu\n", what, a.size(), b.size()); return false; }
    double num = 0, den = 0, mx = 0, mb = 0; int same = 0, rows = (int)(a.size() / V);
    for (size_t i = 0; i < a.size(); ++i) { const double e = (double) a[i] - b[i]; num += e*e; den += (double) b[i]*b[i]; if (fabs(e) > mx) mx = fabs(e); if (fabs(b[i]) > mb) mb = fabs(b[i]); }
    for (int r = 0; r < rows; ++r) { int ia = 0, ib = 0; for (int i = 1; i < V; ++i) { if (a[r*V+i] > a[r*V+ia]) ia = i; if (b[r*V+i] > b[r*V+ib]) ib = i; } same += ia == ib; }
    const double l2 = sqrt(num / (den + 1e-30));
    const bool ok = l2 < tol && std::isfinite(l2);
    printf("%-14s rows %2d  rel-L2 %.3e  max-rel %.3e  argmax equal %d/%d  %s\n", what, rows, l2, mx / (mb + 1e-30), same, rows, ok ? "OK" : "FAIL");
    return ok;
}

int main(int argc, char ** argv) {
    if (argc < 2) { fprintf(stderr, "usage: %s /path/libggml-mi355x.so [workdir]\n", argv[0]); return 2; }
    const std::string work = argc > 2 ? argv[2] : "/tmp";
    llama_backend_init();
    const bool cpu_only = strcmp(argv[1], "cpu") == 0;          // self-check of this program without a GPU: CPU vs CPU
    if (!cpu_only) {
        ggml_backend_reg_t reg = ggml_backend_load(argv[1]);
        if (!reg) { fprintf(stderr, "ggml_backend_load(%s) failed\n", argv[1]); return 2; }
        printf("loaded backend '%s' with %zu device(s)\n", ggml_backend_reg_name(reg), ggml_backend_reg_dev_count(reg));
        if (ggml_backend_reg_dev_count(reg) == 0) { fprintf(stderr, "no MI355X device\n"); return 3; }
    }
    const dims d = { 1024, 8, 4, 2816, 4, 4096, 256 };            // GQA, n_ff not a multiple of 1024, mixed Q4_K / Q6_K like Q4_K_M
    const std::string tp = work + "/dropin_tgt.gguf", dp = work + "/dropin_dft.gguf";
    write_model(tp.c_str(), d, false, GGML_TYPE_Q4_K, 1);
    write_model(dp.c_str(), d, true, GGML_TYPE_Q4_K, 2);
    run_out gpu, cpu;
    if (!run(tp, dp, d, 99, gpu)) return 4;
    if (!run(tp, dp, d, 0, cpu)) return 5;
    if (getenv("DROPIN_TRACE")) {
        const size_t n = std::min(gpu.trace.size(), cpu.trace.size());
        for (size_t i = 0; i < n; ++i) {
            const auto & a = gpu.trace[i].second; const auto & b = cpu.trace[i].second;
            double num = 0, den = 0; for (size_t j = 0; j < std::min(a.size(), b.size()); ++j) { const double e = (double) a[j] - b[j]; num += e*e; den += (double) b[j]*b[j]; }
            printf("trace %3zu %-40s | %-40s n=%zu/%zu relL2 %.3e\n", i, gpu.trace[i].first.c_str(), cpu.trace[i].first.c_str(), a.size(), b.size(), sqrt(num/(den+1e-30)));
        }
    }
    bool ok = true;
    if (getenv("DROPIN_CUT")) {          // the tensors the host read at the cuts: same names in the same order, values as close as the logits
        if (gpu.trace.size() != cpu.trace.size() || gpu.trace.empty()) { printf("cut tensors: %zu vs %zu recorded\n", gpu.trace.size(), cpu.trace.size()); ok = false; }
        double worst = 0; size_t at = 0;
        for (size_t i = 0; i < std::min(gpu.trace.size(), cpu.trace.size()); ++i) {
            const auto & a = gpu.trace[i].second; const auto & b = cpu.trace[i].second;
            if (gpu.trace[i].first != cpu.trace[i].first || a.size() != b.size()) { printf("cut tensor %zu: %s vs %s\n", i, gpu.trace[i].first.c_str(), cpu.trace[i].first.c_str()); ok = false; continue; }
            double num = 0, den = 0; for (size_t j = 0; j < a.size(); ++j) { const double e = (double) a[j] - b[j]; num += e*e; den += (double) b[j]*b[j]; }
            const double l2 = sqrt(num/(den+1e-30)); if (!(l2 <= worst)) { worst = l2; at = i; }
            if (getenv("DROPIN_CUT_VERBOSE")) printf("cut %3zu %-28s n=%zu relL2 %.3e\n", i, gpu.trace[i].first.c_str(), a.size(), l2);
        }
        const bool cut_ok = std::isfinite(worst) && worst < 5e-2;
        printf("%-14s %zu tensors read at the scheduler's cuts, worst rel-L2 %.3e (%s)  %s\n", "cut-readers", gpu.trace.size(), worst, gpu.trace.empty() ? "-" : gpu.trace[at].first.c_str(), cut_ok ? "OK" : "FAIL");
        ok &= cut_ok;
    }
    // tolerance: 5e-2 relative L2 on a 4-layer random model -- see the note on chaotic amplification at the top and
    // tests/test_model_gpu.py, where the bound is calibrated against the reference's own AVX2-vs-scalar spread; the
    // single-layer EAGLE head (no amplification chain) lands at ~5e-7
    ok &= check("prompt", gpu.prompt_logits, cpu.prompt_logits, d.n_vocab, 5e-2);
    ok &= check("step", gpu.step_logits, cpu.step_logits, d.n_vocab, 5e-2);
    ok &= check("tree-verify", gpu.tree_logits, cpu.tree_logits, d.n_vocab, 5e-2);
    ok &= check("batch-of-40", gpu.long_logits, cpu.long_logits, d.n_vocab, 5e-2);
    if (!getenv("DROPIN_CUT")) ok &= check("eagle-draft", gpu.draft_logits, cpu.draft_logits, d.n_vocab, 5e-2);
    printf(ok ? "DROP-IN OK\n" : "DROP-IN FAILED\n");
    remove(tp.c_str()); remove(dp.c_str());
    return ok ? 0 : 1;
}
