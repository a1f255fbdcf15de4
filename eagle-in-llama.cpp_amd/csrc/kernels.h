// kernels.h -- launchers of the hand-written gfx950 kernels.  Every launcher is asynchronous on
// `st`, takes the ggml node whose result it produces (sources are read from node->src[]), and
// assumes mi_supports_op() accepted the node.  Device pointers come from ggml_tensor::data.
#pragma once
#include "mi355x_common.h"

// ---- element-wise / small ops (kernels_ops.hip) ----
void mi_op_rms_norm (hipStream_t st, const ggml_tensor * dst, const ggml_tensor * mul_w /*nullable: fused weight*/, const ggml_tensor * out /*where to write*/);
void mi_op_bin_bcast(hipStream_t st, const ggml_tensor * dst);          // ADD SUB MUL DIV
void mi_op_unary    (hipStream_t st, const ggml_tensor * dst);          // SILU RELU GELU ...
void mi_op_scale    (hipStream_t st, const ggml_tensor * dst);
void mi_op_cpy      (hipStream_t st, const ggml_tensor * src, const ggml_tensor * dst); // CPY CONT DUP
void mi_op_concat   (hipStream_t st, const ggml_tensor * dst);
void mi_op_get_rows (hipStream_t st, const ggml_tensor * dst);
void mi_op_argmax   (hipStream_t st, const ggml_tensor * dst, const ggml_tensor * rows = nullptr);   // rows: the GET_ROWS(table, dst) node fused in
bool mi_argmax_rows_supported(const ggml_tensor * dst, const ggml_tensor * rows);
void mi_op_rope     (hipStream_t st, const ggml_tensor * dst);
void mi_op_soft_max (hipStream_t st, const ggml_tensor * dst);
// k largest entries of n_rows <= 16 rows of an f32 matrix, descending, ties: lower index first; ids / vals are device arrays [n_rows][k]
bool mi_top_k_supported(const ggml_tensor * logits, int k);
void mi_top_k(hipStream_t st, const ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals);
// fused SwiGLU tail: dst = silu(gate) * up   (UNARY(SILU) followed by MUL)
void mi_op_silu_mul (hipStream_t st, const ggml_tensor * gate, const ggml_tensor * up, const ggml_tensor * dst);

// ---- matrix products (kernels_mmvq.hip / kernels_mmf.hip) ----
enum { EPI_F32 = 0, EPI_ROPE_F32 = 1, EPI_ROPE_F16 = 2, EPI_F16 = 3 };
struct act_src {                        // activations of a mat-vec launch (see kernels_mmvq.hip)
    const float * X; int64_t xs; const float * norm_w; int norm; float eps;
    const char * pre;                  // non-NULL: activations already quantised by mi_quant_act (image in HBM scratch)
    const float * X2; int64_t xs2; int ksplit;   // X2 non-NULL: the activations are CONCAT(X, X2) along k, X2 starting at element ksplit (EAGLE's [embd; hidd])
    const float * G; int64_t gs;       // G non-NULL (quantiser launches of the big-batch path only): the activations are silu(G) * X, SwiGLU's product folded into ffn_down's quantiser
    float * norm_out; int64_t norm_os;  // with `norm`: where the folded RMS_NORM [* w] result is materialised as a side effect (row stride in floats); tiled kernel only
};
struct mmvq_mat {
    const char * W; int64_t row_bytes; int rows; int epi;
    char * out; int64_t o_row, o_tok;   // element (row, token) is written at out + row*o_row + token*o_tok  (bytes)
    const float * res; int64_t r_tok;   // optional residual (EPI_F32): res[token*r_tok + row]; r_tok = 0 broadcasts a bias row
    int relu;                           // EPI_F32: max(x, 0) after the residual / bias (fused GGML_UNARY_OP_RELU)
    const int32_t * ids; int n_ids;     // EPI_F32 in the tiled kernel, ids non-NULL: token t's result goes to output row j for every ids[j] == t (fused GET_ROWS(., inp_out_ids))
};
struct mmvq_rope { const int32_t * pos; int head_dim; float theta_scale, freq_scale, attn_factor;
                   const float * tab; };     // tab non-NULL (tiled kernel): {cos, sin} * attn_factor per (token, pair), built once per forward pass by mi_rope_table
// {cos, sin} table of a RoPE (mode NORM) for T positions: [T][head_dim/2][2] floats, the arithmetic of the fused epilogue / ggml_rope_cache_init
void mi_rope_table(hipStream_t st, const int32_t * pos, int T, int head_dim, float theta_scale, float freq_scale, float attn_factor, float * tab);
struct mmvq_launch { act_src act; int k; int n_mat; int swiglu; mmvq_mat m[3]; mmvq_rope rope; int tiled; };   // tiled: every W is in the layout of tile_layout.h (kernels_mmt.hip)
#define MI_ACT_SLOTS 8
struct mi_act_cache {                  // quantised-activation images in HBM scratch (see mi_mmvq_run)
    char * pool = nullptr; size_t slot_bytes = 0; int next = 0; uint64_t epoch = 0;
    struct entry { const void * key; uint64_t epoch; int t0, T, kq, k; } e[MI_ACT_SLOTS] = {};
    // whole-batch images of the big-batch kernel (k_mmt_bb): 4 x 8 MiB, allocated at first use
    char * big_pool = nullptr; int big_next = 0; entry big[4] = {};
    // RoPE table shared by the layers of one forward pass (graph.cpp): valid for `rope_epoch == epoch` and the key below
    float * rope_tab = nullptr; size_t rope_cap = 0 /* floats */; uint64_t rope_epoch = 0; const void * rope_pos = nullptr; int rope_T = 0, rope_hd = 0; float rope_p[3] = {0, 0, 0};
};
int  mi_mmvq_max_tokens(int type, int k);
size_t mi_act_image_bytes(int type, int T, int k);
void mi_quant_act(hipStream_t st, int type, int T, const act_src & a, int k, char * out);
void mi_mmvq_run(hipStream_t st, int type, int n_tokens, const mmvq_launch & L, mi_act_cache * cache, const void * key);
// matrix-core variant for K-quants (kernels_mmq.hip); needs L.act.pre (image written by mi_quant_act)
bool mi_mmq_supported(int type, int T, int k, bool swiglu);
int  mi_mmq_max_tokens(int type, int k, bool swiglu);
bool mi_mmq_inline_quant(int type, int T, const mmvq_launch & L);      // the kernel quantises in-block: no image needed
void mi_mmq_launch(hipStream_t st, int type, int T, const mmvq_launch & L);
// tiled-layout kernel (kernels_mmt.hip): any token count, in-kernel activation quantiser for small T*k
void mi_mmt_run(hipStream_t st, int type, int n_tokens, const mmvq_launch & L, mi_act_cache * cache, const void * key);
// big batches (25+ tokens) as an int8 GEMM on 32x32x32 MFMA tiles (kernels_bb.hip); needs L.act.pre (whole-batch image), one matrix, plain f32 output
bool mi_bb_supported(int type);
void mi_bb_run(hipStream_t st, int type, int n_tokens, const mmvq_launch & L);
// two launches over the same activations with weights of two types (Q4_K | Q5_K, then Q6_K) as one grid; <= 8 tokens, in-kernel quantiser
bool mi_mmt_pair_supported(int typeA, int typeB, int T, const mmvq_launch & LA);
void mi_mmt_run_pair(hipStream_t st, int typeA, int typeB, int T, const mmvq_launch & LA, const mmvq_launch & LB);
void mi_prof_add_bytes(double bytes);                                   // profile hooks: more algorithmic bytes for the launch just recorded
double mi_launch_bytes(const mmvq_launch & L, int T, bool dual);
// HIP-event profile hooks around a mat-vec launch (bench.py roofline): begin returns a record index or -1 when off
int  mi_prof_begin(hipStream_t st, const mmvq_launch & L, int T, bool dual);
void mi_prof_end(hipStream_t st, int idx);
// quantised weight x f32 activations; residual (nullable) is added in the epilogue (fused ADD)
void mi_op_mul_mat_q(hipStream_t st, const ggml_tensor * dst, const ggml_tensor * residual, const ggml_tensor * out, mi_act_cache * cache,
                     const act_src * act = nullptr, const void * key = nullptr);      // act: activation source other than src1 itself (folded norm / SwiGLU product; 2-D operands only)
// f16 / f32 / bf16 src0 x f32 src1 (attention K.q, V.p and unquantised weights)
void mi_op_mul_mat_f(hipStream_t st, const ggml_tensor * dst);
bool mi_mul_mat_q_supported_type(int type);

// ---- fused attention for small batches (kernels_attn.hip) ----
struct mi_attn_args {
    const void * q;      int64_t q_nb1, q_nb2;          // f32 [d, T, H]   (strides in bytes)
    const void * k;      int64_t k_nb1, k_nb2;          // f16 [d, n_kv, H_kv]
    const void * v;      int64_t v_nb1, v_nb2;          // f16 [n_kv, d, H_kv] (transposed V cache: v_nb1 = stride of a head dim), or with
    int v_row;                                          // v_row = 1 (FLASH_ATTN_EXT): f16 [d, n_kv, H_kv], v_nb1 = stride of a cache cell
    const void * mask;   int     mask_f16; int64_t mask_nb1;  // [n_kv, >=T]
    float      * out;    int64_t o_nb1, o_nb2;          // f32, element (dd, h, t) at dd*4 + h*o_nb1 + t*o_nb2
    int d, T, H, H_kv, n_kv;
    float scale;
};
void mi_op_attn_small(hipStream_t st, const mi_attn_args & a);
bool mi_attn_small_supported(const mi_attn_args & a);
