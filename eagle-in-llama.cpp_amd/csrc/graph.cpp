// graph.cpp -- ggml_backend_i::graph_compute for MI355X: walks the nodes the scheduler hands over
// (R/ggml/src/ggml-backend.cpp:1397) and launches hand-written gfx950 kernels, fusing the node
// sequences build_llama / build_eagle emit (R/src/llama.cpp:1647-2019) so that the launch count per
// transformer layer drops from ~25 to ~10:
//     RMS_NORM -> MUL(weight)                  one kernel
//     MUL_MAT(quantised) -> ADD(residual)      residual added in the mat-vec epilogue
//     UNARY(SILU) -> MUL                       one kernel
//     MUL_MAT(K,q) -> SOFT_MAX -> MUL_MAT(V,p) [-> PERMUTE -> CONT]   one attention kernel (small batches)
// Reference counterpart: ggml_cuda_compute_forward (R/ggml/src/ggml-cuda/ggml-cuda.cu:2096) which
// launches one kernel (or more) per node.  supports_op mirrors the role of :2946-3229.
#include "kernels.h"
#include <vector>

static inline bool is_view_op(int op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE;
}
static inline bool is_f32(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32; }

bool mi_supports_op(int, const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE:
            return true;
        case GGML_OP_MUL_MAT: {
            if (!a || !b || !is_f32(b) || !is_f32(op) || b->nb[0] != 4 || op->nb[0] != 4) return false;
            if (b->ne[2] % a->ne[2] || b->ne[3] % a->ne[3]) return false;
            if (mi_mul_mat_q_supported_type(a->type)) {
                const auto tr = mi_traits(a->type);
                if (a->nb[0] != (size_t) tr.size || a->ne[0] % tr.blck) return false;
                if (a->nb[1] != mi_row_size(a->type, a->ne[0])) return false;             // rows must be whole block rows
                if (b->nb[1] % 16) return false;                                          // 16-byte activation loads
                if (a->ne[0] > 128*1024) return false;                                    // one token's int8 image must fit LDS
                return true;
            }
            if (a->type == GGML_TYPE_F32 || a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_BF16)
                return a->nb[0] == (size_t) mi_traits(a->type).size;
            return false;
        }
        case GGML_OP_RMS_NORM:
            return is_f32(a) && is_f32(op) && a->nb[0] == 4 && op->nb[0] == 4;
        case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV:
            return is_f32(a) && is_f32(b) && is_f32(op);
        case GGML_OP_UNARY: {
            if (!is_f32(a) || !is_f32(op) || !mi_is_contiguous(a) || !mi_is_contiguous(op)) return false;
            const int u = mi_op_i32(op, 0);
            return u >= 0 && u < GGML_UNARY_OP_COUNT;
        }
        case GGML_OP_SCALE:
            return is_f32(a) && is_f32(op) && mi_is_contiguous(a) && mi_is_contiguous(op);
        case GGML_OP_CPY: case GGML_OP_CONT: case GGML_OP_DUP: {
            const int s = a->type, d = op->type;
            const bool fs = s == GGML_TYPE_F32 || s == GGML_TYPE_F16, fd = d == GGML_TYPE_F32 || d == GGML_TYPE_F16;
            return (fs && fd) || (s == GGML_TYPE_I32 && d == GGML_TYPE_I32);
        }
        case GGML_OP_CONCAT:
            return is_f32(a) && is_f32(b) && is_f32(op);
        case GGML_OP_GET_ROWS:
            return (a->type == GGML_TYPE_F32 || a->type == GGML_TYPE_F16) && b->type == GGML_TYPE_I32 && is_f32(op);
        case GGML_OP_ROPE: {
            const int mode = mi_op_i32(op, 2);
            if (mode != 0 && mode != GGML_ROPE_TYPE_NEOX) return false;
            return is_f32(a) && is_f32(op) && a->nb[0] == 4 && op->nb[0] == 4 && b && b->type == GGML_TYPE_I32 && (a->ne[0] % 2) == 0;
        }
        case GGML_OP_SOFT_MAX: {
            if (!is_f32(a) || !is_f32(op) || !mi_is_contiguous(a) || !mi_is_contiguous(op)) return false;
            if (b && !((b->type == GGML_TYPE_F32 || b->type == GGML_TYPE_F16) && mi_is_contiguous(b))) return false;
            return true;
        }
        default:
            return false;
    }
}

// ---- use counts (how many later nodes read a tensor), pointer-keyed open addressing
struct use_map {
    std::vector<const ggml_tensor *> key; std::vector<int> cnt; size_t mask;
    explicit use_map(size_t n) { size_t s = 64; while (s < 4*n) s <<= 1; key.assign(s, nullptr); cnt.assign(s, 0); mask = s - 1; }
    size_t slot(const ggml_tensor * t) const { size_t h = ((uintptr_t) t >> 4) * 0x9E3779B97F4A7C15ull; size_t i = (h >> 20) & mask; while (key[i] && key[i] != t) i = (i + 1) & mask; return i; }
    void add(const ggml_tensor * t) { size_t i = slot(t); key[i] = t; cnt[i]++; }
    int  get(const ggml_tensor * t) const { size_t i = slot(t); return key[i] ? cnt[i] : 0; }
};
// a view chain (RESHAPE/VIEW/PERMUTE/TRANSPOSE) resolves to the tensor that owns the data
static const ggml_tensor * view_root(const ggml_tensor * t) { while (t && is_view_op(t->op) && t->src[0] && t->op != GGML_OP_NONE) t = t->src[0]; return t; }

static bool try_fuse_attention(mi_backend_ctx * ctx, ggml_cgraph * g, int i, const use_map & uses, int * consumed);

enum ggml_status mi_graph_compute(mi_backend_ctx * ctx, ggml_cgraph * g) {
    const int n = g->n_nodes;
    use_map uses((size_t) n * 3 + 16);
    for (int i = 0; i < n; ++i) {
        const ggml_tensor * t = g->nodes[i];
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (t->src[s]) uses.add(t->src[s]);
    }
    hipStream_t st = ctx->stream;
    static const bool no_fuse = getenv("GGML_MI355X_NO_FUSION") != nullptr;

    for (int i = 0; i < n; ++i) {
        ggml_tensor * t = g->nodes[i];
        if (is_view_op(t->op) || mi_nelements(t) == 0) continue;
        ggml_tensor * nx = (i + 1 < n) ? g->nodes[i + 1] : nullptr;
        const bool single_use = !no_fuse && uses.get(t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT);

        switch (t->op) {
            case GGML_OP_RMS_NORM: {
                // RMS_NORM -> MUL(norm, weight[ne0]) : llm_build_norm, R/src/llama.cpp:329-365
                if (single_use && nx && nx->op == GGML_OP_MUL && nx->src[0] == t && is_f32(nx->src[1]) && is_f32(nx) &&
                    nx->src[1]->ne[0] == t->ne[0] && mi_nrows(nx->src[1]) == 1 && nx->src[1]->nb[0] == 4 && nx->nb[0] == 4 && mi_same_shape(nx, t)) {
                    mi_op_rms_norm(st, t, nx->src[1], nx); ++i;
                } else {
                    mi_op_rms_norm(st, t, nullptr, t);
                }
            } break;
            case GGML_OP_MUL_MAT: {
                if (mi_mul_mat_q_supported_type(t->src[0]->type)) {
                    // MUL_MAT -> ADD(residual): attention output / ffn_down + inpSA (R/src/llama.cpp:1770,1800)
                    const ggml_tensor * res = nullptr;
                    if (single_use && nx && nx->op == GGML_OP_ADD && is_f32(nx) && nx->nb[0] == 4 && mi_same_shape(nx, t)) {
                        if      (nx->src[0] == t && is_f32(nx->src[1]) && mi_same_shape(nx->src[1], t) && nx->src[1]->nb[0] == 4) res = nx->src[1];
                        else if (nx->src[1] == t && is_f32(nx->src[0]) && mi_same_shape(nx->src[0], t) && nx->src[0]->nb[0] == 4) res = nx->src[0];
                    }
                    if (res) { mi_op_mul_mat_q(st, t, res, nx); ++i; }
                    else     { mi_op_mul_mat_q(st, t, nullptr, t); }
                } else {
                    int consumed = 0;
                    if (!no_fuse && try_fuse_attention(ctx, g, i, uses, &consumed)) { i += consumed; break; }
                    mi_op_mul_mat_f(st, t);
                }
            } break;
            case GGML_OP_UNARY: {
                // SILU(gate) -> MUL(silu, up): llm_build_ffn SwiGLU, R/src/llama.cpp:456-600
                if (single_use && mi_op_i32(t, 0) == GGML_UNARY_OP_SILU && nx && nx->op == GGML_OP_MUL && is_f32(nx) && mi_is_contiguous(nx) && mi_same_shape(nx, t)) {
                    const ggml_tensor * other = nx->src[0] == t ? nx->src[1] : (nx->src[1] == t ? nx->src[0] : nullptr);
                    if (other && is_f32(other) && mi_is_contiguous(other) && mi_same_shape(other, t)) { mi_op_silu_mul(st, t->src[0], other, nx); ++i; break; }
                }
                mi_op_unary(st, t);
            } break;
            case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: mi_op_bin_bcast(st, t); break;
            case GGML_OP_SCALE:    mi_op_scale(st, t); break;
            case GGML_OP_CPY:      mi_op_cpy(st, t->src[0], t->src[1]); break;
            case GGML_OP_CONT: case GGML_OP_DUP: mi_op_cpy(st, t->src[0], t); break;
            case GGML_OP_CONCAT:   mi_op_concat(st, t); break;
            case GGML_OP_GET_ROWS: mi_op_get_rows(st, t); break;
            case GGML_OP_ROPE:     mi_op_rope(st, t); break;
            case GGML_OP_SOFT_MAX: mi_op_soft_max(st, t); break;
            default:
                MI_LOG("graph_compute: op %d (%s) is not supported -- supports_op should have declined it", t->op, t->name);
                return GGML_STATUS_FAILED;
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { MI_LOG("graph_compute: launch error: %s", hipGetErrorString(e)); return GGML_STATUS_FAILED; }
    return GGML_STATUS_SUCCESS;
}

// ---- attention:  kq = MUL_MAT(k, q) ; p = SOFT_MAX(kq, mask, scale) ; kqv = MUL_MAT(v, p) ; [PERMUTE ; CONT]
// as built by llm_build_kqv without flash-attention (R/src/llama.cpp:706-828).
static bool try_fuse_attention(mi_backend_ctx * ctx, ggml_cgraph * g, int i, const use_map & uses, int * consumed) {
    (void) ctx; (void) g; (void) i; (void) uses; (void) consumed;
    return false;   // enabled in kernels_attn.hip once parity-tested
}
