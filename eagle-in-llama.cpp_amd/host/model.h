// model.h -- host-side mirror of the reference's decode path for the llama / eagle architectures:
//   KV-cell bookkeeping           R/src/llama-kv-cache.cpp (find_slot :286-336, seq_rm :368, seq_cp :431, seq_keep :478, cell_max :338)
//   mask / input upload           R/src/llama-context.cpp:61-210 (llama_set_inputs)
//   graph builders                R/src/llama.cpp build_llama :1647, build_eagle :1839, build_lmhead :1813,
//                                 llm_build_{norm :329, fc :367, ffn :456, kv_store :228, kqv :706, kv :830}
//   decode entry points           llama_decode_impl :9486, llama_decode_draft_impl :9978 (hidden-state channel)
// The graphs are emitted node for node in the order ggml_build_forward_expand would give them, with the
// reference's tensor names, and are executed through the backend vtables only (minihost.h).
#pragma once
#include "minihost.h"
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace eh {

enum Ftype { FTYPE_Q4_0 = 0, FTYPE_Q4_K_M = 1, FTYPE_Q8_0 = 2 };

struct ModelConfig {
    int n_embd = 4096, n_head = 32, n_head_kv = 32, head_dim = 128, n_ff = 11008, n_layer = 32, n_vocab = 32000;
    float rms_eps = 1e-6f, rope_base = 10000.0f;
    int n_ctx = 2048;
    int ftype = FTYPE_Q4_K_M;
    bool eagle = false;               // EAGLE head: fc [2*n_embd -> n_embd] + bias + ReLU, no attn_norm / output_norm weights
    int n_seq_max = 16;
    int tp_rank = 0, tp_size = 1;     // tensor parallel: this process holds shard tp_rank of tp_size (one process per GPU)
};

struct Batch {                          // llama_batch: token, pos, seq ids, logits flag (R/include/llama.h:236-253)
    std::vector<int32_t>  token;
    std::vector<int32_t>  pos;
    std::vector<int32_t>  seq_first;    // seq_id[t][0] -- decides the mask row
    std::vector<uint64_t> seq_mask;     // all seq ids of the token as a bit set
    std::vector<uint8_t>  logits;       // output wanted for this token
    std::vector<float>    hidd;         // optional [n_tokens][n_embd] (llama_batch.hidd, llama.h:252)
    int n_tokens() const { return (int) token.size(); }
    void clear() { token.clear(); pos.clear(); seq_first.clear(); seq_mask.clear(); logits.clear(); hidd.clear(); }
    void add(int32_t tok, int32_t p, int32_t seq, bool want_logits) { token.push_back(tok); pos.push_back(p); seq_first.push_back(seq); seq_mask.push_back(1ull << seq); logits.push_back(want_logits); }
};

struct KVCache {
    struct Cell { int32_t pos = -1; uint64_t seqs = 0; };
    uint32_t size = 0, head = 0, n = 0, used = 0;
    std::vector<Cell> cells;
    void init(uint32_t sz) { size = sz; head = 0; n = 0; used = 0; cells.assign(sz, Cell()); }
    bool find_slot(const Batch & b);
    uint32_t cell_max() const;
    void seq_rm(int seq, int32_t p0, int32_t p1);
    void seq_cp(int src, int dst, int32_t p0, int32_t p1);
    void seq_keep(int seq);
    void clear() { for (auto & c : cells) c = Cell(); head = 0; used = 0; }
};

struct Layer { ggml_tensor * attn_norm = nullptr, * wq = nullptr, * wk = nullptr, * wv = nullptr, * wo = nullptr,
               * ffn_norm = nullptr, * gate = nullptr, * up = nullptr, * down = nullptr; };

struct SynthOptions {
    uint64_t seed = 42;
    float accept_p = 0.8f;            // fraction of tokens whose draft embedding predicts the target's next token
    bool  predictable = true;         // structured weights (see model.cpp); false = plain random blocks
};

struct Model {
    ModelConfig cfg;
    mh::Backend * be = nullptr;
    std::unique_ptr<mh::Ctx> wctx;    // weights (+ KV cache tensors)
    std::unique_ptr<mh::Ctx> gctx;    // per-decode graph, buffer re-used between calls
    std::vector<Layer> layers;
    ggml_tensor * output_norm = nullptr, * output = nullptr, * fc = nullptr, * fc_b = nullptr;
    std::vector<uint16_t> tok_embd;   // host-resident f16 [n_vocab][n_embd] (the reference keeps token_embd on the CPU too:
                                      // R/src/llama-model.cpp:1339-1341)
    std::vector<ggml_tensor *> k_l, v_l;
    KVCache kv;
    const Model * lm_head_from = nullptr;   // EAGLE: LM head borrowed from the target (build_lmhead on llm2)
    // tensor parallel: sum a [n_floats] fp32 buffer (device memory of this model's backend) over all ranks, in place,
    // ordered after everything already submitted to the backend (RCCL on the backend stream; gloo in the CPU tests)
    typedef void (*allreduce_fn)(void * user, void * data, int64_t n_floats);
    allreduce_fn allreduce = nullptr; void * allreduce_user = nullptr;
    int n_head_local = 0, n_head_kv_local = 0, n_ff_local = 0;
    int64_t n_allreduce = 0;
    bool tp_in_graph = getenv("EH_TP_SEGMENTS") == nullptr;   // all-reduces enqueued from inside graph_compute (plugin node hooks) when the backend offers them
    static void tp_node_hook(void * user, const ggml_tensor * t, void * stream);
    size_t weight_bytes = 0;          // bytes of all mat-mul weights resident on the device (for the roofline)

    // outputs of the last decode
    mh::HostVec logits;               // [n_outputs][n_vocab]   page-locked: the device writes it with an async copy
    mh::HostVec hidden;               // [n_outputs][n_embd]    (result_norm rows: the hidden-state channel)
    mh::HostVec ids_stage;            // page-locked landing zone of the argmax ids (int32 stored in float slots)
    char * stage_in = nullptr; size_t stage_cap = 0;   // page-locked image of the input tensors of one decode
    std::vector<int32_t> out_ids;     // batch index of each output row
    int n_outputs = 0;
    // greedy fast path: with want_logits = false decode() appends GGML_OP_ARGMAX to the graph and brings back one int per output
    // row (`argmax_ids`) instead of the logits rows; logits stays empty then
    bool want_logits = true;
    // tree drafting: with want_topk = k > 0 (and want_logits) decode() asks the backend for the k best logits of every output row
    // ("ggml_backend_mi355x_top_k") instead of downloading the rows; topk_ids / topk_vals [n_outputs][k] are valid when topk_k > 0,
    // logits stays empty then.  A backend without the extension (reference CPU backend) leaves topk_k = 0 and delivers the logits.
    int want_topk = 0, topk_k = 0;
    std::vector<int32_t> topk_ids; std::vector<float> topk_vals;
    bool topk_ith(int i, const int32_t ** ids, const float ** vals) const;       // by batch index
    std::vector<int32_t> argmax_ids;  // [n_outputs], valid after every decode on the rank that owns the LM head
    int argmax_ith(int i) const;      // by batch index; falls back to scanning logits when they were downloaded
    // timing / stats
    double t_build_us = 0, t_upload_us = 0, t_compute_us = 0, t_download_us = 0; int64_t n_decode = 0;
    int last_n_nodes = 0;
    // teacher forcing (parity tooling): host rows [T][n_embd] that layer il >= 1 of the NEXT decode reads instead of the layer below's output
    std::vector<const float *> forced_layer_inp;
    std::vector<std::pair<ggml_tensor *, const float *>> forced_tensors;

    static Model * create_synthetic(mh::Backend * be, const ModelConfig & cfg, const SynthOptions & opt, const Model * target /* for eagle */);
    ~Model();
    // llama_decode / llama_decode_draft: 0 ok, 1 no KV slot, <0 error (R/src/llama.cpp:9610-9617)
    struct Cut { int node_end; ggml_tensor * t; };          // tensor parallel: partial sums to all-reduce, and where the graph is cut
    int decode(const Batch & batch, bool want_hidden);
    // first half of decode() for a batch whose token ids are not known yet (only its shape is read): KV slots, graph, allocation, masks.
    // A following decode() with the same shape only uploads and launches; any other call drops the preparation (KV cells given back).
    int decode_prepare(const Batch & shape);
    void decode_abandon();
    struct Pending {
        bool valid = false; Batch shape; int T = 0, n_kv = 0; bool tp = false, head_here = true, packed = false, want_logits = true; size_t span = 0;
        std::vector<Cut> cuts; KVCache kv_saved;
        std::vector<const ggml_tensor *> hook_nodes;          // the cut tensors, as handed to the plugin's node hooks
        ggml_tensor * inp_embd = nullptr, * inp_hidd = nullptr, * inp_pos = nullptr, * kq_mask = nullptr, * inp_out = nullptr;
        ggml_tensor * base = nullptr;      // first input tensor: origin of the packed host image
        bool dev_tokens = false;           // the embeddings are fetched on the device (dev_ids below): no inp_embd
        ggml_tensor * result_norm = nullptr, * result_output = nullptr, * result_argmax = nullptr;
    };
    // EAGLE head only, greedy: `n_steps` autoregressive draft steps as ONE graph (device-resident hand-off).  Step 0 is
    // `first` (accepted tokens + target features from the host); step j >= 1 feeds the device-side arg-max token of step j-1
    // (GGML_OP_ARGMAX -> GET_ROWS on a device copy of token_embd) and its result_norm row straight back in.  One upload, one
    // synchronize, n_steps ints come back.  Returns 0 and fills `ids`; 1 when the KV cache has no room (caller falls back).
    // defer_wait: return right after the launch; chain_wait() synchronises and collects the ids (the driver prepares the verification
    // batch in between)
    int decode_chain(const Batch & first, int n_steps, std::vector<int32_t> & ids, bool defer_wait = false);
    int chain_wait(std::vector<int32_t> & ids);
    int chain_steps = 0; double chain_t_launch = 0;
    // Device-side token hand-off of the greedy chain (driver.cpp spec_round): the chain's graph keeps [first token, arg-max of step 0, 1, ...]
    // in ONE i32 tensor; a target whose dev_ids / dev_table point at it (and at the draft's device copy of token_embd) builds its
    // batch embeddings as GET_ROWS(dev_table, dev_ids) -- the verification pass is enqueued right behind the chain, with no host
    // round trip for the drafted tokens in between.  decode() then ignores Batch::token (shape and positions still come from the batch).
    ggml_tensor * chain_ids = nullptr;          // draft: valid from decode_chain() until its next graph build
    // ... and of the verification pass back to the chain: after a decode() with device arg-max the rows of result_norm and the arg-max ids
    // stay where they are in the compute buffer (nothing writes it before this model's next upload, which is stream-ordered behind
    // whatever was enqueued meanwhile).  A draft whose first_* point there reads the re-ingested tokens and features on the device.
    struct DevRows { void * data = nullptr; ggml_backend_buffer_t buffer = nullptr; int rows = 0; };
    DevRows last_norm, last_argmax;             // target: set by decode() (cleared when it fails or runs without the arg-max op)
    DevRows first_feat, first_ids;              // draft: consumed by the next decode_chain() (step 0 inputs), then cleared
    int fetch_last_hidden(float * dst, int rows);   // the same rows for a caller that needs them on the host after all
    const ggml_tensor * dev_ids = nullptr, * dev_table = nullptr;
    const float * logits_ith(int i) const;      // by batch index, like llama_get_logits_ith
    const float * hidden_ith(int i) const;
    size_t matmul_weight_bytes() const { return weight_bytes; }

    Pending pend;
    struct StepIO { ggml_tensor * embd, * hidd, * pos, * mask, * out_ids; int T, n_outputs, n_kv, kv_head; };
    void build_forward(mh::Ctx & g, const StepIO & io, bool tp, std::vector<Cut> * cuts,
                       ggml_tensor *& result_norm, ggml_tensor *& result_output, ggml_tensor *& result_argmax);
    std::unique_ptr<mh::Ctx> ectx;    // device copy of token_embd (f16 [n_embd, n_vocab]), created on the first decode_chain
    ggml_tensor * tok_embd_dev = nullptr;
  public:
    const ggml_tensor * tok_embd_device();      // device copy of token_embd, made on first use (nullptr: no memory)
};

// per-type byte size of a row
size_t row_bytes(int type, int64_t k);
int    weight_type_for(const ModelConfig & cfg, const char * which, int il);

} // namespace eh
