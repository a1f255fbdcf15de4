// graph.cpp -- ggml_backend_i::graph_compute for MI355X: walks the nodes the scheduler hands over
// (R/ggml/src/ggml-backend.cpp:1397) and launches hand-written gfx950 kernels, fusing the node
// sequences build_llama / build_eagle emit (R/src/llama.cpp:1647-2019) so that the launch count per
// transformer layer drops from ~25 to ~10:
//     [RMS_NORM -> MUL(w)] -> {MUL_MAT wq -> ROPE | MUL_MAT wk -> ROPE -> CPY(K cache) | MUL_MAT wv -> CPY(V^T cache)}   1 launch
//     MUL_MAT(K,q) -> SOFT_MAX(tree mask) -> MUL_MAT(V,p) -> PERMUTE -> CONT                                          1 launch
//     MUL_MAT(wo) -> ADD(residual)                                                                                    1 launch
//     [RMS_NORM -> MUL(w)] -> {MUL_MAT gate -> SILU, MUL_MAT up} -> MUL                                               1 launch
//     MUL_MAT(down) -> ADD(residual)                                                                                  1 launch
// i.e. 5 launches per transformer layer instead of ~22 nodes.  A fusion that would move a write earlier than its
// node's position is only taken when the written range cannot alias anything the skipped-over nodes touch
// (ggml-alloc re-uses memory), otherwise the nodes run one by one.
// Reference counterpart: ggml_cuda_compute_forward (R/ggml/src/ggml-cuda/ggml-cuda.cu:2096) which
// launches one kernel (or more) per node.  supports_op mirrors the role of :2946-3229.
#include "kernels.h"
#include <vector>
#include <unordered_set>
#include <algorithm>
#include <cmath>

static inline bool is_view_op(int op) {
    return op == GGML_OP_NONE || op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE;
}
static inline bool is_f32(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32; }

// FLASH_ATTN_EXT(q f32 [d, T, H], k f16 [d, n_kv, H_kv], v f16 [d, n_kv, H_kv], mask f16 [n_kv, pad(T)]) -> f32 [d, H, T]
// (ggml_flash_attn_ext, R/ggml/src/ggml.c; CPU twin ggml_compute_forward_flash_attn_ext_f16): the fused attention kernel with a
// row-major V.  No ALiBi, no logit soft-cap, f16 cache only -- everything else is declined and stays on the CPU.
bool mi_flash_attn_args(const ggml_tensor * op, mi_attn_args & a) {
    const ggml_tensor * q = op->src[0], * k = op->src[1], * v = op->src[2], * mask = op->src[3];
    if (!q || !k || !v || !is_f32(q) || !is_f32(op) || k->type != GGML_TYPE_F16 || v->type != GGML_TYPE_F16) return false;
    if (mi_op_f32(op, 1) != 0.0f || mi_op_f32(op, 2) != 0.0f) return false;                    // max_bias, logit_softcap
    if (q->ne[3] != 1 || k->ne[3] != 1 || v->ne[3] != 1 || q->nb[0] != 4 || k->nb[0] != 2 || v->nb[0] != 2) return false;
    if (k->ne[0] != q->ne[0] || v->ne[0] != q->ne[0] || v->ne[1] != k->ne[1] || v->ne[2] != k->ne[2]) return false;
    if (mask && !(mask->type == GGML_TYPE_F16 && mi_is_contiguous(mask) && mask->ne[0] == k->ne[1] && mask->ne[1] >= q->ne[1] && mask->ne[2] == 1 && mask->ne[3] == 1)) return false;
    if (!mi_is_contiguous(op) || op->ne[0] != v->ne[0] || op->ne[1] != q->ne[2] || op->ne[2] != q->ne[1]) return false;
    a = mi_attn_args{};
    a.d = (int) q->ne[0]; a.T = (int) q->ne[1]; a.H = (int) q->ne[2]; a.n_kv = (int) k->ne[1]; a.H_kv = (int) k->ne[2];
    a.q = q->data; a.q_nb1 = q->nb[1]; a.q_nb2 = q->nb[2];
    a.k = k->data; a.k_nb1 = k->nb[1]; a.k_nb2 = k->nb[2];
    a.v = v->data; a.v_nb1 = v->nb[1]; a.v_nb2 = v->nb[2]; a.v_row = 1;
    a.mask = mask ? mask->data : nullptr; a.mask_f16 = 1; a.mask_nb1 = mask ? mask->nb[1] : 0;
    a.out = (float *) op->data; a.o_nb1 = (int64_t) a.d * 4; a.o_nb2 = (int64_t) a.d * a.H * 4;
    a.scale = mi_op_f32(op, 0);
    // supports_op is also asked before buffers exist (data == NULL): alignment of the base pointers is checked again at run time
    return mi_attn_small_supported(a);
}

bool mi_supports_op(int, const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    // a row-split weight (split.cpp) exists only as per-device slices: nothing but MUL_MAT may read it, and only as src0 (reference: ggml-cuda.cu:2958-2966)
    for (int s = 0; s < GGML_MAX_SRC; ++s) if (op->src[s] && mi_tensor_is_split(op->src[s])) return op->op == GGML_OP_MUL_MAT && s == 0 && mi_split_supports_mul_mat(op);
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
        case GGML_OP_FLASH_ATTN_EXT: {
            mi_attn_args fa;
            return mi_flash_attn_args(op, fa);
        }
        case GGML_OP_ARGMAX:
            return is_f32(a) && a->nb[0] == 4 && a->ne[2] == 1 && a->ne[3] == 1 && op->type == GGML_TYPE_I32 && mi_is_contiguous(op) && a->ne[0] < (1ll << 31);
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

// ---- per-graph side tables, pointer-keyed open addressing
struct tmap {
    std::vector<const ggml_tensor *> key; std::vector<int> a, b; size_t mask;
    explicit tmap(size_t n) { size_t s = 64; while (s < 4*n) s <<= 1; key.assign(s, nullptr); a.assign(s, 0); b.assign(s, -1); mask = s - 1; }
    size_t slot(const ggml_tensor * t) const { size_t h = ((uintptr_t) t >> 4) * 0x9E3779B97F4A7C15ull; size_t i = (h >> 20) & mask; while (key[i] && key[i] != t) i = (i + 1) & mask; return i; }
    size_t touch(const ggml_tensor * t) { size_t i = slot(t); key[i] = t; return i; }
    bool has(const ggml_tensor * t, size_t & i) const { i = slot(t); return key[i] != nullptr; }
};
static inline const ggml_tensor * root_of(const ggml_tensor * t) { return t->view_src ? t->view_src : t; }

struct gctx {
    ggml_cgraph * g; int n;
    tmap uses;        // a = number of nodes that read the tensor directly, b = index of the LAST such node
    tmap rootlast;    // keyed by data owner: b = index of the last node reading it through any view
    tmap index;       // b = node index of a tensor
    std::vector<char> done;
    explicit gctx(ggml_cgraph * gr) : g(gr), n(gr->n_nodes), uses((size_t) gr->n_nodes*3 + 16), rootlast((size_t) gr->n_nodes*3 + 16), index((size_t) gr->n_nodes + 16), done(gr->n_nodes, 0) {
        for (int i = 0; i < n; ++i) {
            const ggml_tensor * t = g->nodes[i];
            index.b[index.touch(t)] = i;
            for (int s = 0; s < GGML_MAX_SRC; ++s) if (t->src[s]) {
                size_t u = uses.touch(t->src[s]); uses.a[u]++; uses.b[u] = i;
                size_t r = rootlast.touch(root_of(t->src[s])); rootlast.b[r] = i;
            }
        }
    }
    int n_uses(const ggml_tensor * t) const { size_t i; return uses.has(t, i) ? uses.a[i] : 0; }
    int last_use(const ggml_tensor * t) const { size_t i; return uses.has(t, i) ? uses.b[i] : -1; }
    int root_last_read(const ggml_tensor * t) const { size_t i; return rootlast.has(root_of(t), i) ? rootlast.b[i] : -1; }
    int idx(const ggml_tensor * t) const { size_t i; return index.has(t, i) ? index.b[i] : -1; }
    // the single node that consumes t, looking through view ops (RESHAPE/PERMUTE/TRANSPOSE/VIEW); nullptr if not unique
    ggml_tensor * sole_consumer(const ggml_tensor * t, bool through_views) const {
        for (int hop = 0; hop < 4; ++hop) {
            if (n_uses(t) != 1 || (t->flags & GGML_TENSOR_FLAG_OUTPUT)) return nullptr;
            ggml_tensor * c = g->nodes[last_use(t)];
            if (through_views && is_view_op(c->op) && c->op != GGML_OP_NONE) { t = c; continue; }
            return c;
        }
        return nullptr;
    }
};
static inline bool overlap(const void * p, size_t n, const void * q, size_t m) { return p && q && (const char *) p < (const char *) q + m && (const char *) q < (const char *) p + n; }
// would writing [p, p+n) at position `at` disturb a node in (at, upto) that still has to run?  (nodes flagged in `skip` do not run)
static bool write_conflicts(const gctx & c, const void * p, size_t n, int at, int upto, const std::vector<char> & skip) {
    for (int j = at + 1; j < upto; ++j) {
        if (skip[j]) continue;
        const ggml_tensor * t = c.g->nodes[j];
        if (is_view_op(t->op)) continue;                              // a view node touches no memory; whoever reads through it is checked by its src
        if (overlap(p, n, t->data, mi_nbytes(t))) return true;
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (t->src[s] && overlap(p, n, t->src[s]->data, mi_nbytes(t->src[s]))) return true;
    }
    return false;
}
static inline bool name_is_result(const ggml_tensor * t) { return strncmp(t->name, "result_", 7) == 0; }

// ---- activation source of a quantised mat-mul: x itself, or RMS_NORM(a) [* w] folded into the prologue
struct act_plan { act_src src; const ggml_tensor * rms = nullptr, * mul = nullptr; };
static bool plan_act(const gctx & c, const ggml_tensor * x, int k, int n_members_using_x, int last_member, act_plan & p) {
    p = act_plan();
    p.src.X = (const float *) x->data; p.src.xs = x->nb[1]/4; p.src.norm = 0; p.src.norm_w = nullptr; p.src.eps = 0; p.src.X2 = nullptr; p.src.xs2 = 0; p.src.ksplit = 0;
    const ggml_tensor * rms = nullptr, * mul = nullptr, * w = nullptr;
    if (x->op == GGML_OP_MUL && x->src[0] && x->src[0]->op == GGML_OP_RMS_NORM) { mul = x; rms = x->src[0]; w = x->src[1]; }
    else if (x->op == GGML_OP_RMS_NORM) rms = x;
    if (x->op == GGML_OP_CONCAT && c.idx(x) >= 0 && c.done[c.idx(x)] == 2) {      // deferred by the CONCAT visit: read both halves in place
        const ggml_tensor * a = x->src[0], * b = x->src[1];
        p.src.X = (const float *) a->data; p.src.xs = a->nb[1]/4;
        p.src.X2 = (const float *) b->data; p.src.xs2 = b->nb[1]/4; p.src.ksplit = (int) a->ne[0];
        return true;
    }
    if (!rms) return true;
    const int ir = c.idx(rms), im = mul ? c.idx(mul) : -1;
    if (ir < 0 || c.done[ir] != 2) return true;                       // 2 = deferred by the RMS_NORM visit below
    (void) im; (void) n_members_using_x; (void) last_member;
    const ggml_tensor * a = rms->src[0];
    p.src.X = (const float *) a->data; p.src.xs = a->nb[1]/4; p.src.norm = 1; p.src.eps = mi_op_f32(rms, 0);
    p.src.norm_w = w ? (const float *) w->data : nullptr;
    p.rms = rms; p.mul = mul;
    (void) k;
    return true;
}
// Can RMS_NORM at node i (optionally followed by MUL with a weight row) be left to its consumers' prologues?
static bool can_defer_norm(const gctx & c, int i) {
    const ggml_tensor * rms = c.g->nodes[i];
    const ggml_tensor * a = rms->src[0];
    if (!is_f32(a) || a->nb[0] != 4 || a->ne[2] != 1 || a->ne[3] != 1 || (a->nb[1] % 16) || ((uintptr_t) a->data % 16)) return false;
    // (more than 24 tokens: every reader is a big-batch launch of its own, and the norm is folded into the ONE quantiser launch that writes
    //  the int8 image they share -- kernels_mmvq.hip k_quant_q8K / k_quant_q80, which also materialise the normalised tensor)
    // a norm the host will read back (graph output, "result_norm": the hidden-state channel) can still be folded when every reader is a
    // TILED launch: those write the normalised tensor as a side effect (kernels_mmt.hip, k_quant_q8K)
    bool wanted = (rms->flags & GGML_TENSOR_FLAG_OUTPUT) || name_is_result(rms);
    const ggml_tensor * x = rms;
    if (c.n_uses(rms) == 1) {
        const ggml_tensor * m = c.g->nodes[c.last_use(rms)];
        if (m->op == GGML_OP_MUL && m->src[0] == rms && c.idx(m) == i + 1) {
            const ggml_tensor * w = m->src[1];
            if (!is_f32(w) || w->ne[0] != rms->ne[0] || mi_nrows(w) != 1 || w->nb[0] != 4 || ((uintptr_t) w->data % 16)) return false;
            if (wanted) return false;                                   // the bare norm is wanted but only norm * w would be written
            wanted = (m->flags & GGML_TENSOR_FLAG_OUTPUT) || name_is_result(m);
            x = m;
        }
    }
    // the first reader of x must be a quantised mat-mul we run with mmvq (x as src1), and `a` must outlive all such readers.  Other
    // readers behind it (the draft chain: the next step's CONCAT reads this step's result_norm) find the tensor materialised by the
    // TILED launches, like the host does for a wanted norm -- so with any of them every mat-mul reader has to be tiled
    const int nu = c.n_uses(x);
    if (nu < 1) return false;
    auto is_mv = [&](const ggml_tensor * t) {
        return t->op == GGML_OP_MUL_MAT && t->src[1] == x && t->src[0] != x && mi_mul_mat_q_supported_type(t->src[0]->type) && t->src[0]->ne[2] == 1 && t->src[0]->ne[3] == 1 &&
               mi_supports_op(0, t) && !mi_tensor_is_split(t->src[0]);              // split weights run on several devices: they need the materialised tensor
    };
    int found = 0, last = -1, n_mv = 0;
    for (int j = c.idx(x) + 1; j < c.n && found < nu; ++j) {
        const ggml_tensor * t = c.g->nodes[j];
        bool reads = false;
        for (int s = 0; s < GGML_MAX_SRC; ++s) if (t->src[s] == x) reads = true;
        if (!reads) continue;
        if (is_mv(t)) { n_mv++; last = j; }
        else { if (n_mv == 0 || is_view_op(t->op)) return false; wanted = true; }        // (a view of x could be read anywhere: not followed)
        found++;
    }
    if (found != nu || n_mv == 0) return false;
    // Every launch that folds a norm writes the normalised tensor as a side effect, wanted here or not: this graph may be one of the
    // scheduler's views, and a reader in the NEXT view (a cut between wq and wk) is invisible from here.  Only tiled launches do
    // that (one block writes, kernels_mmt.hip), so a fold needs every mat-vec reader tiled.
    (void) wanted;
    for (int j = c.idx(x) + 1; j <= last; ++j) { const ggml_tensor * t = c.g->nodes[j]; if (is_mv(t) && !mi_ensure_tiled(t->src[0])) return false; }
    // the materialised tensor must not land on the rows the other blocks of the same launch are still reading (a bare RMS_NORM
    // that ggml-alloc placed in place over its single-use input; with a MUL behind it the scan below catches the MUL at i + 1)
    if (overlap(x->data, mi_nbytes(x), a->data, mi_nbytes(a))) return false;
    if (c.root_last_read(a) <= last) {
        // `a` is not read after the last consumer: its memory may already have been handed to a node in between
        std::vector<char> none(c.n, 0);
        for (int j = i + 1; j <= last; ++j) { const ggml_tensor * t = c.g->nodes[j]; if (overlap(a->data, mi_nbytes(a), t->data, mi_nbytes(t)) && !is_view_op(t->op)) return false; }
    }
    return true;
}

// CONCAT(a, b) along dim 0 at node i feeding exactly one quantised mat-vec at node i+1 (EAGLE's fc over [embd; hidd]): the quantiser
// reads the two halves in place.  Nothing runs between the two nodes, so a and b are still intact when the mat-vec starts.
static bool can_defer_concat(const gctx & c, int i) {
    const ggml_tensor * x = c.g->nodes[i];
    if (mi_op_i32(x, 0) != 0 || c.n_uses(x) != 1 || (x->flags & GGML_TENSOR_FLAG_OUTPUT) || i + 1 >= c.n || x->ne[1] > 24) return false;
    const ggml_tensor * a = x->src[0], * b = x->src[1];
    if (!is_f32(a) || !is_f32(b) || a->nb[0] != 4 || b->nb[0] != 4 || a->ne[1] != b->ne[1] || a->ne[2] != 1 || a->ne[3] != 1 || b->ne[2] != 1 || b->ne[3] != 1) return false;
    if ((a->ne[0] % 4) || (a->nb[1] % 16) || (b->nb[1] % 16) || ((uintptr_t) a->data % 16) || ((uintptr_t) b->data % 16)) return false;
    const ggml_tensor * t = c.g->nodes[i + 1];
    if (t->op != GGML_OP_MUL_MAT || t->src[1] != x || !mi_mul_mat_q_supported_type(t->src[0]->type) || t->src[0]->ne[2] != 1 || t->src[0]->ne[3] != 1) return false;
    if (x->ne[2] != 1 || x->ne[3] != 1 || !mi_supports_op(0, t) || mi_tensor_is_split(t->src[0])) return false;
    return true;
}

// SILU(gate) at node i -> MUL(silu, up) -> quantised mat-mul over >= 25 tokens (ffn_down of a prompt / wide verification batch): the two
// element-wise nodes are left to the quantiser launch of that mat-mul, which reads gate and up in place and never writes the product.
static bool can_defer_swiglu(const gctx & c, int i, int * mul_idx) {
    const ggml_tensor * silu = c.g->nodes[i];
    if (silu->op != GGML_OP_UNARY || mi_op_i32(silu, 0) != GGML_UNARY_OP_SILU || c.n_uses(silu) != 1 || (silu->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    const ggml_tensor * gate = silu->src[0];
    const int im = c.last_use(silu);
    const ggml_tensor * mul = c.g->nodes[im];
    if (mul->op != GGML_OP_MUL || c.n_uses(mul) != 1 || (mul->flags & GGML_TENSOR_FLAG_OUTPUT) || !is_f32(mul) || !mi_is_contiguous(mul)) return false;
    const ggml_tensor * up = mul->src[0] == silu ? mul->src[1] : (mul->src[1] == silu ? mul->src[0] : nullptr);
    if (!up || up == silu || !is_f32(up) || !is_f32(gate) || !mi_is_contiguous(up) || !mi_is_contiguous(gate) || !mi_same_shape(up, mul) || !mi_same_shape(gate, mul)) return false;
    if (mul->ne[2] != 1 || mul->ne[3] != 1 || mul->ne[1] <= 24 || (mul->ne[0] % 256) || ((uintptr_t) up->data % 16) || ((uintptr_t) gate->data % 16)) return false;
    const int ir = c.last_use(mul);
    const ggml_tensor * mm = c.g->nodes[ir];
    if (mm->op != GGML_OP_MUL_MAT || mm->src[1] != mul || mm->src[0] == mul || !mi_mul_mat_q_supported_type(mm->src[0]->type) || mm->src[0]->ne[2] != 1 || mm->src[0]->ne[3] != 1) return false;
    if (!mi_supports_op(0, mm) || mi_tensor_is_split(mm->src[0]) || !mi_ensure_tiled(mm->src[0])) return false;
    // gate dies at the SILU, up at the MUL as far as the allocator knows: nothing that still runs before the mat-mul may write into them,
    // and the mat-mul's own output (or the ADD folded behind it) must not land on them either (token passes re-read them)
    const int iu = c.idx(up);                                          // (up is usually computed AFTER the SILU node: gate, silu, up, mul in DFS order)
    for (int j = i + 1; j <= ir + 1 && j < c.n; ++j) {
        if (j == im) continue;
        const ggml_tensor * t = c.g->nodes[j];
        if (j == ir + 1 && !(t->op == GGML_OP_ADD && (t->src[0] == mm || t->src[1] == mm))) break;
        if (is_view_op(t->op)) continue;
        if (overlap(gate->data, mi_nbytes(gate), t->data, mi_nbytes(t))) return false;
        if (j > iu && overlap(up->data, mi_nbytes(up), t->data, mi_nbytes(t))) return false;
    }
    *mul_idx = im;
    return true;
}

// ---- one member of a multi-matrix mat-vec launch and the nodes its epilogue swallows
struct member {
    int node = -1; const ggml_tensor * mm = nullptr;
    int epi = EPI_F32; const ggml_tensor * out = nullptr;            // tensor that receives the result
    const ggml_tensor * res = nullptr;                                // residual (or broadcast bias row) for EPI_F32
    bool relu = false;                                                // EPI_F32: fused UNARY(RELU)
    const ggml_tensor * ids = nullptr;                                // EPI_F32 with res: out[j] = mm[ids[j]] + res[ids[j]] (the last layer's GET_ROWS pair + ADD)
    bool alt = false;                                                 // weight of the launch's second type (mixed-type pair, kernels_mmt.hip k_mmt2)
    const ggml_tensor * rope = nullptr;
    std::vector<int> swallowed;                                       // node indices done by this member
};
static bool rope_fusable(const ggml_tensor * r, const ggml_tensor * mm) {
    if (r->op != GGML_OP_ROPE || !is_f32(r) || r->src[2]) return false;
    if (mi_op_i32(r, 2) != 0 || mi_op_f32(r, 7) != 0.0f) return false;               // mode NORM, no YaRN mixing
    if (mi_op_i32(r, 1) != r->ne[0] || (r->ne[0] % 2)) return false;                  // n_dims == head size
    if (r->ne[0]*r->ne[1] != mm->ne[0] || r->ne[2] != mm->ne[1] || r->ne[3] != 1) return false;
    if (!mi_is_contiguous(r) || r->src[1]->type != GGML_TYPE_I32) return false;
    return true;
}
// follow mm's result: [RESHAPE] -> ROPE -> (CPY to f16 cache)?  |  [TRANSPOSE] -> CPY f16  |  ADD residual  |  plain
static bool rowsel_fuse_on() { static const bool v = mi_lab_env("GGML_MI355X_NO_ROWSEL_FUSE") == nullptr; return v; }     // A/B: output-row selection / arg-max row fetch fused
static void plan_member(const gctx & c, member & m, bool allow_ids) {
    const ggml_tensor * mm = m.mm;
    m.epi = EPI_F32; m.out = mm; m.res = nullptr; m.relu = false; m.rope = nullptr; m.ids = nullptr; m.swallowed.clear();
    if ((mm->flags & GGML_TENSOR_FLAG_OUTPUT) || c.n_uses(mm) != 1) return;
    const ggml_tensor * c1 = c.g->nodes[c.last_use(mm)];
    // the last layer keeps only the rows that produce outputs (R/src/llama.cpp build_llama / build_eagle: cur = get_rows(cur, inp_out_ids);
    // inpSA = get_rows(inpSA, inp_out_ids); ffn_inp = add(cur, inpSA)): three nodes right behind the output projection, done by its epilogue
    if (allow_ids && rowsel_fuse_on() && c1->op == GGML_OP_GET_ROWS && c1->src[0] == mm && c.idx(c1) == m.node + 1 && m.node + 3 < c.n && c.n_uses(c1) == 1 && !(c1->flags & GGML_TENSOR_FLAG_OUTPUT)) {
        const ggml_tensor * ids = c1->src[1], * g2 = c.g->nodes[m.node + 2], * ad = c.g->nodes[m.node + 3];
        const ggml_tensor * other = g2->src[0];
        if (ids->type == GGML_TYPE_I32 && ids->ne[1] == 1 && ids->ne[2] == 1 && ids->ne[3] == 1 && ids->nb[0] == 4 && ids->ne[0] <= mm->ne[1] &&
            g2->op == GGML_OP_GET_ROWS && g2->src[1] == ids && c.n_uses(g2) == 1 && !(g2->flags & GGML_TENSOR_FLAG_OUTPUT) &&
            other && is_f32(other) && other->nb[0] == 4 && mi_same_shape(other, mm) && other != mm &&
            ad->op == GGML_OP_ADD && ((ad->src[0] == c1 && ad->src[1] == g2) || (ad->src[0] == g2 && ad->src[1] == c1)) &&
            is_f32(ad) && ad->nb[0] == 4 && mi_same_shape(ad, c1) && mi_same_shape(g2, c1) && c1->ne[0] == mm->ne[0] && c1->ne[2] == 1 && c1->ne[3] == 1 &&
            !overlap(ad->data, mi_nbytes(ad), other->data, mi_nbytes(other))) {
            m.res = other; m.out = ad; m.ids = ids;
            m.swallowed.push_back(m.node + 1); m.swallowed.push_back(m.node + 2); m.swallowed.push_back(m.node + 3);
            return;
        }
    }
    // optional reshape
    const ggml_tensor * v = c1; std::vector<int> sw;
    if (v->op == GGML_OP_RESHAPE && v->src[0] == mm && c.n_uses(v) == 1 && !(v->flags & GGML_TENSOR_FLAG_OUTPUT)) { sw.push_back(c.idx(v)); v = c.g->nodes[c.last_use(v)]; }
    if (v->op == GGML_OP_ROPE && rope_fusable(v, mm) && (v->src[0] == mm || (sw.size() && v->src[0] == c.g->nodes[sw.back()]))) {
        const ggml_tensor * rope = v;
        // K: rope -> CPY into an f16 view (contiguous token-major slice of the cache)
        if (c.n_uses(rope) == 1 && !(rope->flags & GGML_TENSOR_FLAG_OUTPUT)) {
            const ggml_tensor * cp = c.g->nodes[c.last_use(rope)];
            if (cp->op == GGML_OP_CPY && cp->src[0] == rope && cp->src[1]->type == GGML_TYPE_F16 && mi_is_contiguous(cp->src[1]) &&
                mi_nelements(cp->src[1]) == mi_nelements(rope)) {
                m.epi = EPI_ROPE_F16; m.out = cp->src[1]; m.rope = rope; m.swallowed = sw; m.swallowed.push_back(c.idx(rope)); m.swallowed.push_back(c.idx(cp));
                return;
            }
        }
        m.epi = EPI_ROPE_F32; m.out = rope; m.rope = rope; m.swallowed = sw; m.swallowed.push_back(c.idx(rope));
        return;
    }
    // V: [TRANSPOSE] -> CPY into an f16 [T, rows] view with arbitrary row stride
    if (c1->op == GGML_OP_TRANSPOSE && c1->src[0] == mm && c.n_uses(c1) == 1) {
        const ggml_tensor * cp = c.g->nodes[c.last_use(c1)];
        if (cp->op == GGML_OP_CPY && cp->src[0] == c1 && cp->src[1]->type == GGML_TYPE_F16 && cp->src[1]->ne[0] == mm->ne[1] && cp->src[1]->ne[1] == mm->ne[0] &&
            cp->src[1]->ne[2] == 1 && cp->src[1]->ne[3] == 1 && cp->src[1]->nb[0] == 2) {
            m.epi = EPI_F16; m.out = cp->src[1]; m.swallowed.push_back(c.idx(c1)); m.swallowed.push_back(c.idx(cp));
            return;
        }
    }
    // residual: the very next node adds a same-shaped f32 tensor -- or a bias row broadcast over the tokens (EAGLE's fc.bias)
    const ggml_tensor * tail = mm;
    if (c1->op == GGML_OP_ADD && c.idx(c1) == m.node + 1 && is_f32(c1) && c1->nb[0] == 4 && mi_same_shape(c1, mm)) {
        const ggml_tensor * other = c1->src[0] == mm ? c1->src[1] : c1->src[0];
        const bool bias = other != mm && c1->src[0] == mm && is_f32(other) && other->nb[0] == 4 && other->ne[0] == mm->ne[0] && mi_nrows(other) == 1;
        if (other != mm && is_f32(other) && other->nb[0] == 4 && (mi_same_shape(other, mm) || bias)) { m.res = other; m.out = c1; m.swallowed.push_back(c.idx(c1)); tail = c1; }
    }
    // RELU right behind (fc -> +bias -> relu)
    if (c.n_uses(tail) == 1 && !(tail->flags & GGML_TENSOR_FLAG_OUTPUT)) {
        const ggml_tensor * u = c.g->nodes[c.last_use(tail)];
        if (u->op == GGML_OP_UNARY && mi_op_i32(u, 0) == GGML_UNARY_OP_RELU && u->src[0] == tail && c.idx(u) == c.idx(tail) + 1 && is_f32(u) && mi_is_contiguous(u) && mi_same_shape(u, mm) && u->nb[0] == 4) {
            m.relu = true; m.out = u; m.swallowed.push_back(c.idx(u));
        }
    }
}
static void fill_mat(mmvq_mat & M, const member & m) {
    const ggml_tensor * w = m.mm->src[0];
    M.W = (const char *) w->data; M.row_bytes = w->nb[1]; M.rows = (int) w->ne[1]; M.epi = m.epi; M.res = nullptr; M.r_tok = 0; M.relu = m.relu ? 1 : 0; M.ids = nullptr; M.n_ids = 0;
    M.out = (char *) m.out->data;
    switch (m.epi) {
        case EPI_F32:      M.o_row = 4; M.o_tok = m.out->nb[1]; if (m.res) { M.res = (const float *) m.res->data; M.r_tok = mi_nrows(m.res) == 1 ? 0 : m.res->nb[1]/4; }
                           if (m.ids) { M.ids = (const int32_t *) m.ids->data; M.n_ids = (int) m.ids->ne[0]; } break;
        case EPI_ROPE_F32: M.o_row = 4; M.o_tok = (int64_t) w->ne[1] * 4; break;                         // rope out is contiguous [d, heads, T]
        case EPI_ROPE_F16: M.o_row = 2; M.o_tok = (int64_t) w->ne[1] * 2; break;                         // contiguous f16 slice
        case EPI_F16:      M.o_row = m.out->nb[1]; M.o_tok = 2; break;                                   // out [T, rows]: token fastest
    }
}

// Every layer of a forward pass rotates by the same positions: the {cos, sin} * attn_factor values for (token, pair) are computed once per
// graph_compute (a 4 us launch; theta by the reference's float recurrence) instead of in every epilogue of every layer (up to 63 dependent
// multiplies, then sinf / cosf: ~4 us per q|k|v launch at 6 tokens).  Built when the graph holds at least `min_users` ROPE nodes over these
// positions and the root position tensor has at most `max_tokens` entries; returns the rows for `pos` (which may be a slice of a longer
// tensor: the draft chain keeps the positions of all its steps in one input and hands every step a view) or NULL.
static const float * rope_table_rows(mi_backend_ctx * ctx, const gctx & c, hipStream_t st, const mmvq_rope & R, const ggml_tensor * pos, int T, int64_t max_tokens, int min_users) {
    static const bool tab_on = mi_lab_env("GGML_MI355X_NO_ROPE_TABLE") == nullptr;
    mi_act_cache * ac = ctx->act_cache;
    const ggml_tensor * root = pos->view_src ? pos->view_src : pos;
    const int64_t poff = ((const char *) pos->data - (const char *) root->data) / 4, Troot = root->ne[0];
    if (!(tab_on && Troot <= max_tokens && R.head_dim <= 256 && (R.head_dim % 2) == 0 && pos->type == GGML_TYPE_I32 && root->type == GGML_TYPE_I32 &&
          root->nb[0] == 4 && mi_nrows(root) == 1 && poff >= 0 && poff + T <= Troot)) return nullptr;
    const bool hit = ac->rope_tab && ac->rope_epoch == ac->epoch && ac->rope_pos == root->data && ac->rope_T == (int) Troot && ac->rope_hd == R.head_dim &&
                     ac->rope_p[0] == R.theta_scale && ac->rope_p[1] == R.freq_scale && ac->rope_p[2] == R.attn_factor;
    if (!hit) {
        int users = 0;
        for (int j = 0; j < c.n; ++j) { const ggml_tensor * r = c.g->nodes[j]; if (r->op == GGML_OP_ROPE && r->src[1] && (r->src[1] == root || r->src[1]->view_src == root)) users++; }
        if (users < min_users) return nullptr;
        const size_t need = (size_t) Troot * (R.head_dim / 2) * 2;
        if (need > ac->rope_cap) {                                     // grows rarely (first prompt): launches already queued may still read the old table
            if (ac->rope_tab) { HIP_CHECK(hipStreamSynchronize(st)); HIP_CHECK(hipFree(ac->rope_tab)); ac->rope_tab = nullptr; }
            ac->rope_cap = need < (size_t) 24 * 128 * 2 ? (size_t) 24 * 128 * 2 : need + need / 2;
            HIP_CHECK(hipMalloc((void **) &ac->rope_tab, ac->rope_cap * sizeof(float)));
        }
        mi_rope_table(st, (const int32_t *) root->data, (int) Troot, R.head_dim, R.theta_scale, R.freq_scale, R.attn_factor, ac->rope_tab);
        ac->rope_epoch = ac->epoch; ac->rope_pos = root->data; ac->rope_T = (int) Troot; ac->rope_hd = R.head_dim;
        ac->rope_p[0] = R.theta_scale; ac->rope_p[1] = R.freq_scale; ac->rope_p[2] = R.attn_factor;
    }
    return ac->rope_tab + (size_t) poff * (R.head_dim / 2) * 2;
}

// try to run MUL_MAT node i together with its siblings; returns true when handled
static bool run_mmvq_group(mi_backend_ctx * ctx, gctx & c, int i, bool fuse) {
    ggml_tensor * t = c.g->nodes[i];
    const ggml_tensor * w0 = t->src[0], * x = t->src[1];
    hipStream_t st = ctx->stream;
    const bool simple2d = w0->ne[2] == 1 && w0->ne[3] == 1 && x->ne[2] == 1 && x->ne[3] == 1;
    // more than 24 tokens (prompts, wide tree verification): every matrix runs on its own through the one-pass big-batch kernel
    // (kernels_mmt.hip, k_mmt_bb: weights streamed once for the whole batch); only the residual ADD is still folded in
    const bool big = x->ne[1] > 24;
    if (!fuse || !simple2d || big) {
        // RMS_NORM deferred to us?  only possible when fuse is on, so nothing to undo here
        const ggml_tensor * res = nullptr; ggml_tensor * nx = (i + 1 < c.n) ? c.g->nodes[i + 1] : nullptr;
        if (fuse && nx && nx->op == GGML_OP_ADD && c.n_uses(t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT) && is_f32(nx) && nx->nb[0] == 4 && mi_same_shape(nx, t)) {
            if      (nx->src[0] == t && is_f32(nx->src[1]) && mi_same_shape(nx->src[1], t) && nx->src[1]->nb[0] == 4) res = nx->src[1];
            else if (nx->src[1] == t && is_f32(nx->src[0]) && mi_same_shape(nx->src[0], t) && nx->src[0]->nb[0] == 4) res = nx->src[0];
        }
        // activation source other than x itself: a norm or a SwiGLU product left to this mat-mul's quantiser launch (big batches, fuse on)
        act_src as{}; const act_src * asp = nullptr; const void * key = nullptr;
        if (fuse && simple2d && big) {
            act_plan ap; plan_act(c, x, (int) w0->ne[0], 1, i, ap);
            const int ix = c.idx(x);
            if (ap.rms) {
                MI_ASSERT(mi_ensure_tiled(t->src[0]));
                as = ap.src; as.norm_out = (float *) x->data; as.norm_os = x->nb[1]/4; asp = &as; key = ap.rms;
                if (res && overlap(nx->data, mi_nbytes(nx), x->data, mi_nbytes(x))) as.norm_out = nullptr;      // that memory has been handed on: nobody reads the norm any more
            } else if (x->op == GGML_OP_MUL && ix >= 0 && c.done[ix] == 2) {                                    // can_defer_swiglu
                const ggml_tensor * silu = (x->src[0]->op == GGML_OP_UNARY && c.idx(x->src[0]) >= 0 && c.done[c.idx(x->src[0])] == 2) ? x->src[0] : x->src[1];
                const ggml_tensor * up = x->src[0] == silu ? x->src[1] : x->src[0];
                as = ap.src; as.X = (const float *) up->data; as.xs = up->nb[1]/4; as.G = (const float *) silu->src[0]->data; as.gs = silu->src[0]->nb[1]/4;
                asp = &as; key = x;
            }
        }
        // epilogues of the one-pass GEMM (kernels_bb.hip): RoPE by table (q), RoPE -> f16 K-cache slice (k), transposed f16 V-cache store (v) --
        // the four element-wise launches per layer a big batch used to carry behind q | k | v
        if (fuse && simple2d && big && !res && mi_bb_supported(w0->type) && mi_ensure_tiled(t->src[0])) {
            member m; m.node = i; m.mm = t; plan_member(c, m, false);
            if (m.out != m.mm && (m.epi == EPI_F16 || m.epi == EPI_ROPE_F32 || m.epi == EPI_ROPE_F16)) {
                const int T = (int) x->ne[1];
                mmvq_launch L{};
                if (asp) L.act = *asp; else { L.act.X = (const float *) x->data; L.act.xs = x->nb[1]/4; }
                L.k = (int) w0->ne[0]; L.n_mat = 1; L.tiled = 1;
                bool ok = true;
                if (m.rope) {
                    const ggml_tensor * pos = m.rope->src[1];
                    L.rope.pos = (const int32_t *) pos->data; L.rope.head_dim = (int) m.rope->ne[0];
                    L.rope.theta_scale = powf(mi_op_f32(m.rope, 5), -2.0f / mi_op_i32(m.rope, 1));
                    L.rope.freq_scale = mi_op_f32(m.rope, 6); L.rope.attn_factor = mi_op_f32(m.rope, 8);
                    L.rope.tab = rope_table_rows(ctx, c, st, L.rope, pos, T, 16384, 1);
                    ok = L.rope.tab != nullptr && (w0->ne[1] % 2) == 0;
                }
                // the tail's buffer is written here instead of at its own node: nothing in between may still need that memory, and it must not
                // be the activations' (a later token pass re-reads them)
                std::vector<char> sk(c.n, 0); sk[i] = 1; for (int sidx : m.swallowed) sk[sidx] = 1;
                const int orig = *std::max_element(m.swallowed.begin(), m.swallowed.end());
                const size_t nb = mi_nbytes(m.out);
                if (write_conflicts(c, m.out->data, nb, i, orig, sk) || overlap(m.out->data, nb, L.act.X, (size_t) T * L.act.xs * 4) ||
                    (L.act.G && overlap(m.out->data, nb, L.act.G, (size_t) T * L.act.gs * 4)) || (L.act.norm_out && overlap(m.out->data, nb, x->data, mi_nbytes(x)))) ok = false;
                if (ok) {
                    fill_mat(L.m[0], m);
                    mi_mmvq_run(st, w0->type, T, L, ctx->act_cache, key ? key : (const void *) x);
                    for (int sidx : m.swallowed) c.done[sidx] = 1;
                    return true;
                }
            }
        }
        if (res) { mi_op_mul_mat_q(st, t, res, nx, ctx->act_cache, asp, key); c.done[i + 1] = 1; } else mi_op_mul_mat_q(st, t, nullptr, t, ctx->act_cache, asp, key);
        return true;
    }
    const int k = (int) w0->ne[0], T = (int) x->ne[1];
    // ---- collect siblings: later MUL_MATs with the same src1, same weight type
    member mem[4]; int nm = 0, nalt = 0;
    mem[nm].node = i; mem[nm].mm = t; nm++;
    const bool tiled0 = mi_ensure_tiled(t->src[0]);                       // first use re-lays the weight out (tile_layout.h); a launch is all-tiled or all row-major
    // siblings of a second weight type (Q4_K_M / Q5_K_M: wv is Q6_K beside Q4_K|Q5_K wq, wk) ride in the same grid: ONE member at most
    int alt_type = -1;
    { mmvq_launch probe{}; probe.k = k; probe.act.norm = 1; if (tiled0 && mi_mmt_pair_supported(w0->type, GGML_TYPE_Q6_K, T, probe)) alt_type = GGML_TYPE_Q6_K; }
    for (int j = i + 1; j < c.n && j < i + 14 && nm < 3; ++j) {
        const ggml_tensor * u = c.g->nodes[j];
        if (c.done[j] || u->op != GGML_OP_MUL_MAT || u->src[1] != x) continue;
        const bool alt = u->src[0]->type != w0->type;
        if (alt && (nalt || (int) u->src[0]->type != alt_type)) continue;
        if (u->src[0]->ne[0] != k || u->src[0]->ne[2] != 1 || u->src[0]->ne[3] != 1 || !mi_supports_op(0, u)) continue;
        if (mi_ensure_tiled(u->src[0]) != tiled0) continue;
        mem[nm].node = j; mem[nm].mm = u; mem[nm].alt = alt; nm++; nalt += alt;
    }
    // ---- SwiGLU: gate (this node) -> SILU -> MUL(silu, up)
    if (nm >= 2 && c.n_uses(t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT)) {
        const ggml_tensor * silu = c.g->nodes[c.last_use(t)];
        if (silu->op == GGML_OP_UNARY && mi_op_i32(silu, 0) == GGML_UNARY_OP_SILU && c.n_uses(silu) == 1 && !(silu->flags & GGML_TENSOR_FLAG_OUTPUT)) {
            const ggml_tensor * mul = c.g->nodes[c.last_use(silu)];
            for (int q = 1; q < nm; ++q) {
                const ggml_tensor * up = mem[q].mm;
                if (mem[q].alt) continue;
                if (mul->op == GGML_OP_MUL && ((mul->src[0] == silu && mul->src[1] == up) || (mul->src[1] == silu && mul->src[0] == up)) &&
                    c.n_uses(up) == 1 && !(up->flags & GGML_TENSOR_FLAG_OUTPUT) && mi_same_shape(up, t) && is_f32(mul) && mi_is_contiguous(mul) &&
                    up->src[0]->ne[1] == w0->ne[1]) {
                    act_plan ap; plan_act(c, x, k, 2, mem[q].node, ap);
                    std::vector<char> skip(c.n, 0);
                    skip[i] = skip[c.idx(silu)] = skip[mem[q].node] = skip[c.idx(mul)] = 1;
                    const int at = i, upto = c.idx(mul);
                    // the product is written at `at` instead of `upto`: must not alias anything still needed, nor our own inputs
                    if (write_conflicts(c, mul->data, mi_nbytes(mul), at, upto, skip)) break;
                    if (overlap(mul->data, mi_nbytes(mul), ap.src.X, (size_t) T * ap.src.xs * 4)) break;
                    mmvq_launch L{}; L.act = ap.src; L.k = k; L.n_mat = 2; L.swiglu = 1; L.tiled = tiled0;
                    MI_ASSERT(tiled0 || !ap.rms);
                    if (ap.rms && !overlap(mul->data, mi_nbytes(mul), x->data, mi_nbytes(x))) { L.act.norm_out = (float *) x->data; L.act.norm_os = x->nb[1]/4; }      // the folded norm is materialised as a side effect
                    member g0 = mem[0], g1 = mem[q]; g0.epi = EPI_F32; g0.out = mul; g0.res = nullptr; g1.epi = EPI_F32; g1.out = mul; g1.res = nullptr;
                    fill_mat(L.m[0], g0); fill_mat(L.m[1], g1);
                    L.m[0].o_row = 4; L.m[0].o_tok = mul->nb[1];
                    mi_mmvq_run(st, w0->type, T, L, ctx->act_cache, ap.rms ? (const void *) ap.rms : (const void *) x);
                    c.done[i] = c.done[c.idx(silu)] = c.done[mem[q].node] = c.done[c.idx(mul)] = 1;
                    return true;
                }
            }
        }
    }
    // ---- generic group: per-member epilogues; hoisted members must be alias-safe
    act_plan ap; plan_act(c, x, k, nm, mem[nm - 1].node, ap);
    std::vector<char> skip(c.n, 0);
    int keep = 0;
    member sel[3];
    const ggml_tensor * pos = nullptr; const ggml_tensor * rope0 = nullptr;
    for (int q = 0; q < nm; ++q) {
        member m = mem[q];
        plan_member(c, m, tiled0 && T <= 8); m.alt = mem[q].alt;      // (row selection in the epilogue: single-pass tiled launches only)
        if (m.rope) {     // all ropes of a launch must share positions and parameters
            if (rope0 && (m.rope->src[1] != rope0->src[1] || memcmp(m.rope->op_params, rope0->op_params, sizeof(int32_t)*11) != 0)) { if (q == 0) {} m.epi = EPI_F32; m.out = m.mm; m.rope = nullptr; m.swallowed.clear(); }
            else { rope0 = m.rope; pos = m.rope->src[1]; }
        }
        if (q == 0 && m.out != m.mm) {
            // the first member's fused tail writes a LATER node's buffer while other blocks still read the activations: ggml-alloc may have
            // given that node the activations' own memory (free once this node has run) -- then the tail stays unfused
            const size_t nb = mi_nbytes(m.out);
            if (overlap(m.out->data, nb, ap.src.X, (size_t) T * ap.src.xs * 4) || (ap.src.X2 && overlap(m.out->data, nb, ap.src.X2, (size_t) T * ap.src.xs2 * 4))) {
                m.epi = EPI_F32; m.out = m.mm; m.res = nullptr; m.relu = false; m.rope = nullptr; m.ids = nullptr; m.swallowed.clear();
            }
        }
        if (q > 0) {
            // hoisting: this member's write moves from its own position (or its CPY's) up to node i
            std::vector<char> sk = skip; sk[m.node] = 1; for (int s : m.swallowed) sk[s] = 1;
            const int orig = m.swallowed.empty() ? m.node : std::max(m.node, *std::max_element(m.swallowed.begin(), m.swallowed.end()));
            const size_t nb = mi_nbytes(m.out);
            static const bool dbg = mi_lab_env("GGML_MI355X_DEBUG_GROUP") != nullptr;
            if (write_conflicts(c, m.out->data, nb, i, orig, sk)) { if (dbg) MI_LOG("group at %s: member %s not hoisted (write conflict, epi %d, out %s)", t->name, m.mm->name, m.epi, m.out->name); continue; }   // leave it to run at its own position
            if (overlap(m.out->data, nb, ap.src.X, (size_t) T * ap.src.xs * 4) || (ap.src.X2 && overlap(m.out->data, nb, ap.src.X2, (size_t) T * ap.src.xs2 * 4))) { if (dbg) MI_LOG("group at %s: member %s not hoisted (overlaps activations)", t->name, m.mm->name); continue; }
        }
        skip[m.node] = 1; for (int s : m.swallowed) skip[s] = 1;
        sel[keep++] = m;
    }
    if (keep == 0) return false;
    if (ap.src.X2) {       // deferred CONCAT: ggml-alloc may have handed the memory of its sources to one of our outputs -- then build it after all
        bool clash = false;
        for (int q = 0; q < keep; ++q) {
            const size_t nb = mi_nbytes(sel[q].out);
            if (overlap(sel[q].out->data, nb, ap.src.X, (size_t) T * ap.src.xs * 4) || overlap(sel[q].out->data, nb, ap.src.X2, (size_t) T * ap.src.xs2 * 4)) clash = true;
        }
        if (clash) { mi_op_concat(st, x); ap.src.X = (const float *) x->data; ap.src.xs = x->nb[1]/4; ap.src.X2 = nullptr; ap.src.xs2 = 0; ap.src.ksplit = 0; }
    }
    { static const bool dbg = mi_lab_env("GGML_MI355X_DEBUG_GROUP") != nullptr; if (dbg) MI_LOG("group at %s: %d siblings, %d kept, T=%d", t->name, nm, keep, T); }
    mmvq_launch L{}; L.act = ap.src; L.k = k; L.n_mat = 0; L.swiglu = 0; L.tiled = tiled0;
    if (tiled0 && ap.rms) { L.act.norm_out = (float *) x->data; L.act.norm_os = x->nb[1]/4; }
    if (rope0) {
        L.rope.pos = (const int32_t *) pos->data; L.rope.head_dim = (int) rope0->ne[0];
        L.rope.theta_scale = powf(mi_op_f32(rope0, 5), -2.0f / mi_op_i32(rope0, 1));
        L.rope.freq_scale = mi_op_f32(rope0, 6); L.rope.attn_factor = mi_op_f32(rope0, 8);
        // a chain step's single layer keeps the in-epilogue form (fewer than four ROPE nodes over these positions)
        L.rope.tab = (tiled0 && T <= 24) ? rope_table_rows(ctx, c, st, L.rope, pos, T, 24, 4) : nullptr;
    }
    MI_ASSERT(tiled0 || !ap.rms);                                          // can_defer_norm: a fold needs tiled readers
    // a member whose (hoisted) output shares memory with the norm tensor: that memory has been handed on, nobody reads the norm any more
    for (int q = 0; q < keep; ++q) if (L.act.norm_out && sel[q].out && overlap(sel[q].out->data, mi_nbytes(sel[q].out), x->data, mi_nbytes(x))) L.act.norm_out = nullptr;
    mmvq_launch LB = L; LB.act.norm_out = nullptr;                         // the second type's partition of a mixed-type launch
    for (int q = 0; q < keep; ++q) { mmvq_launch & D = sel[q].alt ? LB : L; fill_mat(D.m[D.n_mat++], sel[q]); }
    const void * key = ap.rms ? (const void *) ap.rms : (const void *) x;
    if (LB.n_mat && L.n_mat && mi_mmt_pair_supported(w0->type, alt_type, T, L)) mi_mmt_run_pair(st, w0->type, alt_type, T, L, LB);
    else {       // (the hoisted second type as a launch of its own when the activation source does not suit the in-kernel quantiser)
        if (L.n_mat)  mi_mmvq_run(st, w0->type, T, L, ctx->act_cache, key);
        if (LB.n_mat) { LB.act.norm_out = L.n_mat ? nullptr : L.act.norm_out; mi_mmvq_run(st, alt_type, T, LB, ctx->act_cache, key); }
    }
    for (int q = 0; q < keep; ++q) { c.done[sel[q].node] = 1; for (int s : sel[q].swallowed) c.done[s] = 1; }
    return true;
}

// ---- attention:  kq = MUL_MAT(k, q) ; p = SOFT_MAX(kq, mask, scale) ; kqv = MUL_MAT(v, p) ; PERMUTE ; CONT
static bool run_attention(mi_backend_ctx * ctx, gctx & c, int i) {
    const ggml_tensor * kq = c.g->nodes[i];
    const ggml_tensor * k = kq->src[0], * q = kq->src[1];
    if (k->type != GGML_TYPE_F16 || !is_f32(q) || k->nb[0] != 2 || q->nb[0] != 4 || k->ne[3] != 1 || q->ne[3] != 1) return false;
    if (c.n_uses(kq) != 1 || (kq->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    const ggml_tensor * sm = c.g->nodes[c.last_use(kq)];
    if (sm->op != GGML_OP_SOFT_MAX || sm->src[0] != kq || mi_op_f32(sm, 1) != 0.0f || c.n_uses(sm) != 1 || (sm->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    const ggml_tensor * mask = sm->src[1];
    if (mask && !((mask->type == GGML_TYPE_F32 || mask->type == GGML_TYPE_F16) && mi_is_contiguous(mask) && mask->ne[0] == kq->ne[0] && mask->ne[1] >= kq->ne[1])) return false;
    const ggml_tensor * kqv = c.g->nodes[c.last_use(sm)];
    if (kqv->op != GGML_OP_MUL_MAT || kqv->src[1] != sm || c.n_uses(kqv) != 1 || (kqv->flags & GGML_TENSOR_FLAG_OUTPUT)) return false;
    const ggml_tensor * v = kqv->src[0];
    if (v->type != GGML_TYPE_F16 || v->nb[0] != 2 || v->ne[3] != 1 || v->ne[0] != k->ne[1] || v->ne[1] != k->ne[0] || v->ne[2] != k->ne[2]) return false;
    const ggml_tensor * pm = c.g->nodes[c.last_use(kqv)];
    if (pm->op != GGML_OP_PERMUTE || pm->src[0] != kqv || c.n_uses(pm) != 1) return false;
    if (!(mi_op_i32(pm, 0) == 0 && mi_op_i32(pm, 1) == 2 && mi_op_i32(pm, 2) == 1 && mi_op_i32(pm, 3) == 3)) return false;
    const ggml_tensor * ct = c.g->nodes[c.last_use(pm)];
    if (ct->op != GGML_OP_CONT || ct->src[0] != pm || !is_f32(ct) || !mi_is_contiguous(ct)) return false;
    mi_attn_args a{};
    a.d = (int) k->ne[0]; a.n_kv = (int) k->ne[1]; a.H_kv = (int) k->ne[2]; a.T = (int) q->ne[1]; a.H = (int) q->ne[2];
    if (kq->ne[0] != a.n_kv || kq->ne[1] != a.T || kq->ne[2] != a.H || mi_nelements(ct) != (int64_t) a.d * a.H * a.T) return false;
    a.q = q->data; a.q_nb1 = q->nb[1]; a.q_nb2 = q->nb[2];
    a.k = k->data; a.k_nb1 = k->nb[1]; a.k_nb2 = k->nb[2];
    a.v = v->data; a.v_nb1 = v->nb[1]; a.v_nb2 = v->nb[2];
    a.mask = mask ? mask->data : nullptr; a.mask_f16 = mask && mask->type == GGML_TYPE_F16; a.mask_nb1 = mask ? mask->nb[1] : 0;
    a.out = (float *) ct->data; a.o_nb1 = (int64_t) a.d * 4; a.o_nb2 = (int64_t) a.d * a.H * 4;
    a.scale = mi_op_f32(sm, 0);
    if (!mi_attn_small_supported(a)) return false;
    // the result lands at node i instead of at CONT's position
    std::vector<char> skip(c.n, 0);
    const int ism = c.idx(sm), ikqv = c.idx(kqv), ipm = c.idx(pm), ict = c.idx(ct);
    skip[i] = skip[ism] = skip[ikqv] = skip[ipm] = skip[ict] = 1;
    if (write_conflicts(c, ct->data, mi_nbytes(ct), i, ict, skip)) return false;
    if (overlap(ct->data, mi_nbytes(ct), q->data, mi_nbytes(q))) return false;
    mi_op_attn_small(ctx->stream, a);
    c.done[i] = c.done[ism] = c.done[ikqv] = c.done[ipm] = c.done[ict] = 1;
    return true;
}

enum ggml_status mi_graph_compute(mi_backend_ctx * ctx, ggml_cgraph * g) {
    gctx c(g);
    const int n = c.n;
    ctx->act_cache->epoch++;
    hipStream_t st = ctx->stream;
    static const bool no_fuse = getenv("GGML_MI355X_NO_FUSION") != nullptr;
    const bool fuse = !no_fuse;

    // node hooks (backend.cpp: ggml_backend_mi355x_set_node_hooks): fired when the loop passes the producing node -- its work (or the
    // fused launch that swallowed it, earlier) is queued by then
    std::unordered_set<const ggml_tensor *> hooked;
    if (ctx->node_hook) for (int j = 0; j < ctx->n_hook_nodes; ++j) hooked.insert(ctx->hook_nodes[j]);
    auto fire = [&](const ggml_tensor * t) { if (!hooked.empty() && hooked.count(t)) ctx->node_hook(ctx->node_hook_user, t, (void *) st); };

    for (int i = 0; i < n; ++i) {
        ggml_tensor * t = g->nodes[i];
        if (c.done[i] || is_view_op(t->op) || mi_nelements(t) == 0) { if (c.done[i] != 2) fire(t); continue; }
        ggml_tensor * nx = (i + 1 < n) ? g->nodes[i + 1] : nullptr;
        const bool single_use = fuse && c.n_uses(t) == 1 && !(t->flags & GGML_TENSOR_FLAG_OUTPUT);
        { static const bool trace = mi_lab_env("GGML_MI355X_TRACE_OPS") != nullptr; if (trace) MI_LOG("node %4d op %2d %-24s [%lld %lld %lld] src0 %s src1 %s", i, (int) t->op, t->name, (long long) t->ne[0], (long long) t->ne[1], (long long) t->ne[2], t->src[0] ? t->src[0]->name : "-", t->src[1] ? t->src[1]->name : "-"); }

        switch (t->op) {
            case GGML_OP_RMS_NORM: {
                if (fuse && can_defer_norm(c, i)) {                   // folded into the prologue of the mat-vecs that read it
                    c.done[i] = 2;
                    if (c.n_uses(t) == 1 && nx && nx->op == GGML_OP_MUL && nx->src[0] == t) c.done[i + 1] = 2;
                    break;
                }
                if (single_use && nx && nx->op == GGML_OP_MUL && nx->src[0] == t && is_f32(nx->src[1]) && is_f32(nx) &&
                    nx->src[1]->ne[0] == t->ne[0] && mi_nrows(nx->src[1]) == 1 && nx->src[1]->nb[0] == 4 && nx->nb[0] == 4 && mi_same_shape(nx, t)) {
                    mi_op_rms_norm(st, t, nx->src[1], nx); c.done[i + 1] = 1;
                } else {
                    mi_op_rms_norm(st, t, nullptr, t);
                }
            } break;
            case GGML_OP_MUL_MAT: {
                if (mi_tensor_is_split(t->src[0])) { mi_split_mul_mat(ctx, t); break; }                 // -sm row: slices on every device, gathered here
                if (mi_mul_mat_q_supported_type(t->src[0]->type)) {
                    if (t->src[0]->view_src && mi_is_tiled(t->src[0]->view_src)) mi_untile(t->src[0], true);      // a view into a re-laid-out weight: back to ggml's layout for good
                    if (!run_mmvq_group(ctx, c, i, fuse)) mi_op_mul_mat_q(st, t, nullptr, t, ctx->act_cache);
                } else {
                    if (fuse && run_attention(ctx, c, i)) break;
                    mi_op_mul_mat_f(st, t);
                }
            } break;
            case GGML_OP_UNARY: {
                { int im = -1; if (fuse && can_defer_swiglu(c, i, &im)) { c.done[i] = 2; c.done[im] = 2; break; } }
                if (single_use && mi_op_i32(t, 0) == GGML_UNARY_OP_SILU && nx && nx->op == GGML_OP_MUL && is_f32(nx) && mi_is_contiguous(nx) && mi_same_shape(nx, t)) {
                    const ggml_tensor * other = nx->src[0] == t ? nx->src[1] : (nx->src[1] == t ? nx->src[0] : nullptr);
                    if (other && is_f32(other) && mi_is_contiguous(other) && mi_same_shape(other, t)) { mi_op_silu_mul(st, t->src[0], other, nx); c.done[i + 1] = 1; break; }
                }
                mi_op_unary(st, t);
            } break;
            case GGML_OP_ADD: case GGML_OP_SUB: case GGML_OP_MUL: case GGML_OP_DIV: mi_op_bin_bcast(st, t); break;
            case GGML_OP_SCALE:    mi_op_scale(st, t); break;
            case GGML_OP_CPY:      mi_op_cpy(st, t->src[0], t->src[1]); break;
            case GGML_OP_CONT: case GGML_OP_DUP: mi_op_cpy(st, t->src[0], t); break;
            case GGML_OP_CONCAT:   if (fuse && can_defer_concat(c, i)) { c.done[i] = 2; break; } mi_op_concat(st, t); break;
            case GGML_OP_GET_ROWS: mi_op_get_rows(st, t); break;
            case GGML_OP_ARGMAX: {
                // GET_ROWS(table, this arg-max) as the very next node (the draft chain's token -> embedding hand-off): one launch
                const ggml_tensor * gr = (fuse && rowsel_fuse_on() && i + 1 < c.n && !c.done[i + 1]) ? c.g->nodes[i + 1] : nullptr;
                if (gr && gr->op == GGML_OP_GET_ROWS && mi_argmax_rows_supported(t, gr) && !overlap(gr->data, mi_nbytes(gr), t->src[0]->data, mi_nbytes(t->src[0])) &&
                    !overlap(gr->data, mi_nbytes(gr), t->data, mi_nbytes(t))) { mi_op_argmax(st, t, gr); c.done[i + 1] = 1; }
                else mi_op_argmax(st, t);
            } break;
            case GGML_OP_FLASH_ATTN_EXT: { mi_attn_args fa; if (!mi_flash_attn_args(t, fa)) { MI_LOG("flash_attn_ext: operands changed since supports_op"); return GGML_STATUS_FAILED; } mi_op_attn_small(st, fa); } break;
            case GGML_OP_ROPE:     mi_op_rope(st, t); break;
            case GGML_OP_SOFT_MAX: mi_op_soft_max(st, t); break;
            default:
                MI_LOG("graph_compute: op %d (%s) is not supported -- supports_op should have declined it", t->op, t->name);
                return GGML_STATUS_FAILED;
        }
        if (c.done[i] != 2) fire(t);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { MI_LOG("graph_compute: launch error: %s", hipGetErrorString(e)); return GGML_STATUS_FAILED; }
    mi_tile_release_scratch(ctx->device);                             // first-use weight conversions of this graph are done
    return GGML_STATUS_SUCCESS;
}
