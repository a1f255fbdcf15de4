// minihost.cpp -- see minihost.h
#include "minihost.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>

namespace mh {

type_traits traits(int type) {
    switch (type) {
        case GGML_TYPE_F32:  return {1, 4};   case GGML_TYPE_F16:  return {1, 2};   case GGML_TYPE_BF16: return {1, 2};
        case GGML_TYPE_I8:   return {1, 1};   case GGML_TYPE_I16:  return {1, 2};   case GGML_TYPE_I32:  return {1, 4};
        case GGML_TYPE_I64:  return {1, 8};   case GGML_TYPE_F64:  return {1, 8};
        case GGML_TYPE_Q4_0: return {32, 18}; case GGML_TYPE_Q4_1: return {32, 20}; case GGML_TYPE_Q5_0: return {32, 22};
        case GGML_TYPE_Q5_1: return {32, 24}; case GGML_TYPE_Q8_0: return {32, 34}; case GGML_TYPE_Q8_1: return {32, 36};
        case GGML_TYPE_Q2_K: return {256, 84};  case GGML_TYPE_Q3_K: return {256, 110}; case GGML_TYPE_Q4_K: return {256, 144};
        case GGML_TYPE_Q5_K: return {256, 176}; case GGML_TYPE_Q6_K: return {256, 210}; case GGML_TYPE_Q8_K: return {256, 292};
        default: return {0, 0};
    }
}
size_t row_size(int type, int64_t ne0) { auto t = traits(type); return (size_t)(ne0 / t.blck) * t.size; }
int64_t nelements(const ggml_tensor * t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }
size_t nbytes(const ggml_tensor * t) {
    auto tr = traits(t->type);
    for (int i = 0; i < 4; ++i) if (t->ne[i] <= 0) return 0;
    size_t n;
    if (tr.blck == 1) { n = tr.size; for (int i = 0; i < 4; ++i) n += (t->ne[i] - 1) * t->nb[i]; }
    else { n = t->ne[0] * t->nb[0] / tr.blck; for (int i = 1; i < 4; ++i) n += (t->ne[i] - 1) * t->nb[i]; }
    return n;
}
bool is_contiguous(const ggml_tensor * t) {
    auto tr = traits(t->type);
    size_t next = tr.size;
    if (t->ne[0] != tr.blck && t->nb[0] != next) return false;
    next *= t->ne[0] / tr.blck;
    for (int i = 1; i < 4; ++i) { if (t->ne[i] != 1) { if (t->nb[i] != next) return false; next *= t->ne[i]; } else next = t->ne[i]*next; }
    return true;
}

// ------------------------------------------------------------------ Backend
Backend * Backend::load(const char * so_path, const char * entry_symbol, int device_index, std::string * err) {
    void * dl = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
    if (!dl) { if (err) *err = std::string("dlopen failed: ") + dlerror(); return nullptr; }
    typedef ggml_backend_reg_t (*init_t)(void);
    init_t init = (init_t) dlsym(dl, entry_symbol);
    if (!init) { if (err) *err = std::string("symbol not found: ") + entry_symbol; dlclose(dl); return nullptr; }
    ggml_backend_reg_t reg = init();
    if (!reg || reg->api_version != GGML_BACKEND_API_VERSION) { if (err) *err = "backend registry missing or api_version != 1"; return nullptr; }
    const size_t ndev = reg->iface.get_device_count(reg);
    if ((size_t) device_index >= ndev) { if (err) *err = "no such device (count=" + std::to_string(ndev) + ")"; return nullptr; }
    Backend * b = new Backend;
    b->dl = dl; b->reg = reg; b->path = so_path; b->entry = entry_symbol;
    b->dev = reg->iface.get_device(reg, device_index);
    b->be  = b->dev->iface.init_backend(b->dev, nullptr);
    if (!b->be) { if (err) *err = "init_backend failed"; delete b; return nullptr; }
    b->buft = b->dev->iface.get_buffer_type(b->dev);
    b->is_host = b->buft->iface.is_host && b->buft->iface.is_host(b->buft);
    return b;
}
Backend::~Backend() { if (be) be->iface.free(be); /* the library stays mapped: its statics own reg/dev objects */ }
const char * Backend::name() const { return be->iface.get_name(be); }
void Backend::set_n_threads(int n) {
    if (!reg->iface.get_proc_address) return;
    typedef void (*fn_t)(ggml_backend_t, int);
    fn_t fn = (fn_t) reg->iface.get_proc_address(reg, "ggml_backend_set_n_threads");
    if (fn) fn(be, n);
}
void Backend::synchronize() { if (be->iface.synchronize) be->iface.synchronize(be); }
bool Backend::supports_op(const ggml_tensor * t) const { return dev->iface.supports_op(dev, t); }
ggml_backend_buffer_t Backend::alloc_buffer(size_t size, int usage) {
    ggml_backend_buffer_t b = buft->iface.alloc_buffer(buft, size);
    if (b) b->usage = (enum ggml_backend_buffer_usage) usage;
    return b;
}
void * Backend::host_alloc(size_t size) {
    ggml_backend_buffer_type_t hb = dev->iface.get_host_buffer_type ? dev->iface.get_host_buffer_type(dev) : nullptr;
    if (hb) {
        ggml_backend_buffer_t b = hb->iface.alloc_buffer(hb, size);
        if (b) { void * p = b->iface.get_base(b); host_blocks.push_back({ p, b }); return p; }
    }
    void * p = malloc(size ? size : 1);
    host_blocks.push_back({ p, nullptr });
    return p;
}
void Backend::host_free(void * p) {
    for (size_t i = 0; i < host_blocks.size(); ++i) if (host_blocks[i].p == p) {
        if (host_blocks[i].buf) free_buffer(host_blocks[i].buf); else free(p);
        host_blocks.erase(host_blocks.begin() + i);
        return;
    }
}
void Backend::free_buffer(ggml_backend_buffer_t b) {
    // same sequence as ggml_backend_buffer_free (R/ggml/src/ggml-backend.cpp): backend hook, then the object
    if (!b) return;
    if (b->iface.free_buffer) b->iface.free_buffer(b);
    delete b;
}

// ------------------------------------------------------------------ Ctx
Ctx::~Ctx() { for (auto b : buffers) be->free_buffer(b); }

ggml_tensor * Ctx::new_tensor(int type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3, const char * name) {
    ggml_tensor * t = pool.push();
    memset(t, 0, sizeof(*t));
    auto tr = traits(type);
    t->type = (enum ggml_type) type;
    t->ne[0] = ne0; t->ne[1] = ne1; t->ne[2] = ne2; t->ne[3] = ne3;
    t->nb[0] = tr.size; t->nb[1] = t->nb[0] * (ne0 / tr.blck);
    t->nb[2] = t->nb[1] * ne1; t->nb[3] = t->nb[2] * ne2;
    t->op = GGML_OP_NONE;
    if (name) set_name(t, name);
    return t;
}
void Ctx::set_name(ggml_tensor * t, const char * name) { snprintf(t->name, sizeof(t->name), "%s", name); }

ggml_tensor * Ctx::view_impl(ggml_tensor * a, int op) {
    ggml_tensor * r = new_tensor(a->type, a->ne[0], a->ne[1], a->ne[2], a->ne[3]);
    for (int i = 0; i < 4; ++i) r->nb[i] = a->nb[i];
    r->op = (enum ggml_op) op; r->src[0] = a;
    r->view_src = a->view_src ? a->view_src : a;
    r->view_offs = a->view_src ? a->view_offs : 0;
    if (a->data) { r->data = a->data; r->buffer = a->buffer; }
    nodes.push_back(r);
    return r;
}
ggml_tensor * Ctx::view(ggml_tensor * a, int n_dims, const int64_t * ne, const size_t * nb, size_t offset) {
    ggml_tensor * r = view_impl(a, GGML_OP_VIEW);
    for (int i = 0; i < 4; ++i) r->ne[i] = i < n_dims ? ne[i] : 1;
    r->nb[0] = a->nb[0];
    // ggml_view_{2,3,4}d: given strides for the leading dims, trailing ones continue contiguously
    for (int i = 1; i < 4; ++i) r->nb[i] = (i < n_dims) ? nb[i] : r->nb[i-1] * r->ne[i-1];
    if (n_dims == 1) { r->nb[1] = r->nb[0] * (r->ne[0] / traits(a->type).blck); r->nb[2] = r->nb[1]; r->nb[3] = r->nb[2]; }
    r->view_offs += offset;
    memcpy(r->op_params, &offset, sizeof(offset));
    if (a->data) r->data = (char *) a->data + offset;
    return r;
}
ggml_tensor * Ctx::view_1d(ggml_tensor * a, int64_t ne0, size_t offset) { int64_t ne[4] = {ne0,1,1,1}; size_t nb[4] = {0,0,0,0}; return view(a, 1, ne, nb, offset); }
ggml_tensor * Ctx::view_2d(ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset) { int64_t ne[4] = {ne0,ne1,1,1}; size_t nb[4] = {0,nb1,0,0}; return view(a, 2, ne, nb, offset); }
ggml_tensor * Ctx::view_3d(ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset) {
    int64_t ne[4] = {ne0,ne1,ne2,1}; size_t nb[4] = {0,nb1,nb2,0}; return view(a, 3, ne, nb, offset);
}
ggml_tensor * Ctx::reshape(ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) {
    ggml_tensor * r = view_impl(a, GGML_OP_RESHAPE);
    auto tr = traits(a->type);
    r->ne[0] = ne0; r->ne[1] = ne1; r->ne[2] = ne2; r->ne[3] = ne3;
    r->nb[0] = tr.size; r->nb[1] = r->nb[0] * (ne0 / tr.blck); r->nb[2] = r->nb[1]*ne1; r->nb[3] = r->nb[2]*ne2;
    return r;
}
ggml_tensor * Ctx::permute(ggml_tensor * a, int ax0, int ax1, int ax2, int ax3) {
    ggml_tensor * r = view_impl(a, GGML_OP_PERMUTE);
    const int ax[4] = {ax0, ax1, ax2, ax3};
    for (int i = 0; i < 4; ++i) { r->ne[ax[i]] = a->ne[i]; r->nb[ax[i]] = a->nb[i]; }
    for (int i = 0; i < 4; ++i) r->op_params[i] = ax[i];
    return r;
}
ggml_tensor * Ctx::transpose(ggml_tensor * a) {
    ggml_tensor * r = view_impl(a, GGML_OP_TRANSPOSE);
    r->ne[0] = a->ne[1]; r->ne[1] = a->ne[0]; r->nb[0] = a->nb[1]; r->nb[1] = a->nb[0];
    return r;
}
ggml_tensor * Ctx::op_result(int op, int type, const int64_t * ne, ggml_tensor * a, ggml_tensor * b, ggml_tensor * c) {
    ggml_tensor * r = new_tensor(type, ne[0], ne[1], ne[2], ne[3]);
    r->op = (enum ggml_op) op; r->src[0] = a; r->src[1] = b; r->src[2] = c;
    nodes.push_back(r);
    return r;
}
ggml_tensor * Ctx::cont(ggml_tensor * a) { return op_result(GGML_OP_CONT, a->type, a->ne, a); }
ggml_tensor * Ctx::cont_2d(ggml_tensor * a, int64_t ne0, int64_t ne1) { int64_t ne[4] = {ne0, ne1, 1, 1}; return op_result(GGML_OP_CONT, a->type, ne, a); }
ggml_tensor * Ctx::cpy(ggml_tensor * a, ggml_tensor * b) {
    ggml_tensor * r = view_impl(b, GGML_OP_CPY);      // result is a view of the destination
    r->src[0] = a; r->src[1] = b;
    return r;
}
ggml_tensor * Ctx::mul_mat(ggml_tensor * a, ggml_tensor * b) { int64_t ne[4] = {a->ne[1], b->ne[1], b->ne[2], b->ne[3]}; return op_result(GGML_OP_MUL_MAT, GGML_TYPE_F32, ne, a, b); }
ggml_tensor * Ctx::rms_norm(ggml_tensor * a, float eps) { ggml_tensor * r = op_result(GGML_OP_RMS_NORM, a->type, a->ne, a); memcpy(&r->op_params[0], &eps, 4); return r; }
ggml_tensor * Ctx::bin(int op, ggml_tensor * a, ggml_tensor * b) { return op_result(op, a->type, a->ne, a, b); }
ggml_tensor * Ctx::add(ggml_tensor * a, ggml_tensor * b) { return bin(GGML_OP_ADD, a, b); }
ggml_tensor * Ctx::mul(ggml_tensor * a, ggml_tensor * b) { return bin(GGML_OP_MUL, a, b); }
ggml_tensor * Ctx::unary(ggml_tensor * a, int uop) { ggml_tensor * r = op_result(GGML_OP_UNARY, a->type, a->ne, a); r->op_params[0] = uop; return r; }
ggml_tensor * Ctx::scale(ggml_tensor * a, float s) { ggml_tensor * r = op_result(GGML_OP_SCALE, a->type, a->ne, a); memcpy(&r->op_params[0], &s, 4); return r; }
ggml_tensor * Ctx::concat(ggml_tensor * a, ggml_tensor * b, int dim) {
    int64_t ne[4]; for (int i = 0; i < 4; ++i) ne[i] = (i == dim) ? a->ne[i] + b->ne[i] : a->ne[i];
    ggml_tensor * r = op_result(GGML_OP_CONCAT, a->type, ne, a, b); r->op_params[0] = dim; return r;
}
ggml_tensor * Ctx::get_rows(ggml_tensor * a, ggml_tensor * b) { int64_t ne[4] = {a->ne[0], b->ne[0], b->ne[1], b->ne[2]}; return op_result(GGML_OP_GET_ROWS, GGML_TYPE_F32, ne, a, b); }
ggml_tensor * Ctx::argmax(ggml_tensor * a) { int64_t ne[4] = {a->ne[1], 1, 1, 1}; return op_result(GGML_OP_ARGMAX, GGML_TYPE_I32, ne, a); }
ggml_tensor * Ctx::rope_ext(ggml_tensor * a, ggml_tensor * pos, ggml_tensor * ff, int n_dims, int mode, int n_ctx_orig,
                            float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow) {
    ggml_tensor * r = op_result(GGML_OP_ROPE, a->type, a->ne, a, pos, ff);
    int32_t p[15] = {0, n_dims, mode, 0, n_ctx_orig};
    memcpy(p + 5, &freq_base, 4); memcpy(p + 6, &freq_scale, 4); memcpy(p + 7, &ext_factor, 4);
    memcpy(p + 8, &attn_factor, 4); memcpy(p + 9, &beta_fast, 4); memcpy(p + 10, &beta_slow, 4);
    p[11] = p[12] = p[13] = p[14] = 0;
    memcpy(r->op_params, p, sizeof(p));
    return r;
}
ggml_tensor * Ctx::soft_max_ext(ggml_tensor * a, ggml_tensor * mask, float scale, float max_bias) {
    ggml_tensor * r = op_result(GGML_OP_SOFT_MAX, a->type, a->ne, a, mask);
    memcpy(&r->op_params[0], &scale, 4); memcpy(&r->op_params[1], &max_bias, 4);
    return r;
}

bool Backend::top_k(const ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals) {
    typedef int (*fn_t)(ggml_backend_t, const ggml_tensor *, const int32_t *, int, int, int32_t *, float *);
    static const char * name = "ggml_backend_mi355x_top_k";
    if (!reg || !reg->iface.get_proc_address) return false;
    fn_t fn = (fn_t) reg->iface.get_proc_address(reg, name);
    return fn && fn(be, logits, rows, n_rows, k, ids, vals) == 0;
}
bool Backend::set_node_hooks(const ggml_tensor * const * nodes, int n, node_hook_fn fn, void * user) {
    typedef int (*fn_t)(ggml_backend_t, const ggml_tensor * const *, int, node_hook_fn, void *);
    if (!reg || !reg->iface.get_proc_address) return false;
    fn_t f = (fn_t) reg->iface.get_proc_address(reg, "ggml_backend_mi355x_set_node_hooks");
    return f && f(be, nodes, n, fn, user) == 0;
}
bool Ctx::use_split(int main_device, const float * tensor_split) {
    if (!be->reg->iface.get_proc_address) return false;
    typedef ggml_backend_buffer_type_t (*fn_t)(int, const float *);
    fn_t fn = (fn_t) be->reg->iface.get_proc_address(be->reg, "ggml_backend_split_buffer_type");
    if (!fn) return false;
    buft_override = fn(main_device, tensor_split);
    return buft_override != nullptr;
}
bool Ctx::alloc() {
    ggml_backend_buffer_type_t bt = buft_override ? buft_override : be->buft;
    const size_t align = bt->iface.get_alignment(bt);
    auto asize = [&](const ggml_tensor * t) -> size_t {
        size_t s = bt->iface.get_alloc_size ? bt->iface.get_alloc_size(bt, t) : nbytes(t);
        return (s + align - 1) / align * align;
    };
    size_t total = 0;
    for (auto & t : pool) if (!t.data && !t.view_src) total += asize(&t);
    if (total) {
        ggml_backend_buffer_t buf = nullptr;
        if (reuse && !buffers.empty() && reuse_cap >= total) buf = buffers.back();
        else {
            if (reuse && !buffers.empty()) { be->synchronize(); be->free_buffer(buffers.back()); buffers.pop_back(); }
            const size_t cap = reuse ? total + total/4 + (1u << 20) : total;
            if (buft_override) { buf = bt->iface.alloc_buffer(bt, cap); if (buf) buf->usage = (enum ggml_backend_buffer_usage) usage; }
            else buf = be->alloc_buffer(cap, usage);
            if (!buf) return false;
            buffers.push_back(buf);
            if (reuse) reuse_cap = cap;
        }
        char * base = (char *) buf->iface.get_base(buf);
        size_t off = 0;
        for (auto & t : pool) if (!t.data && !t.view_src) {
            t.data = base + off; t.buffer = buf; off += asize(&t);
            if (buf->iface.init_tensor) buf->iface.init_tensor(buf, &t);
        }
    }
    for (auto & t : pool) if (t.view_src && !t.data) {           // views resolve after their roots
        t.data = (char *) t.view_src->data + t.view_offs; t.buffer = t.view_src->buffer;
    }
    return true;
}
void Ctx::reset_graph() { pool.clear(); nodes.clear(); reuse = true; }
void Ctx::set(ggml_tensor * t, const void * data, size_t offset, size_t size) { t->buffer->iface.set_tensor(t->buffer, t, data, offset, size); }
void Ctx::get(const ggml_tensor * t, void * data, size_t offset, size_t size) { t->buffer->iface.get_tensor(t->buffer, t, data, offset, size); }
void Ctx::set_async(ggml_tensor * t, const void * data, size_t offset, size_t size) {
    if (be->be->iface.set_tensor_async) be->be->iface.set_tensor_async(be->be, t, data, offset, size); else set(t, data, offset, size);
}
void Ctx::get_async(const ggml_tensor * t, void * data, size_t offset, size_t size) {
    if (be->be->iface.get_tensor_async) be->be->iface.get_tensor_async(be->be, t, data, offset, size); else get(t, data, offset, size);
}
enum ggml_status Ctx::compute_async() {
    node_array.assign(nodes.begin(), nodes.end());
    memset(&graph, 0, sizeof(graph));
    graph.size = (int) node_array.size(); graph.n_nodes = (int) node_array.size(); graph.n_leafs = 0;
    graph.nodes = node_array.data(); graph.order = GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT;
    return be->be->iface.graph_compute(be->be, &graph);
}
enum ggml_status Ctx::compute_range(int n0, int n1) {
    if (n1 <= n0) return GGML_STATUS_SUCCESS;
    node_array.assign(nodes.begin() + n0, nodes.begin() + n1);
    memset(&graph, 0, sizeof(graph));
    graph.size = (int) node_array.size(); graph.n_nodes = (int) node_array.size(); graph.n_leafs = 0;
    graph.nodes = node_array.data(); graph.order = GGML_CGRAPH_EVAL_ORDER_LEFT_TO_RIGHT;
    return be->be->iface.graph_compute(be->be, &graph);
}
enum ggml_status Ctx::compute() {
    const auto t0 = std::chrono::steady_clock::now();
    enum ggml_status s = compute_async();
    const auto t1 = std::chrono::steady_clock::now();
    be->synchronize();
    const auto t2 = std::chrono::steady_clock::now();
    t_issue_us += std::chrono::duration<double, std::micro>(t1 - t0).count();
    t_wait_us  += std::chrono::duration<double, std::micro>(t2 - t1).count();
    return s;
}

} // namespace mh
