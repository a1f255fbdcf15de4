// capi.cpp -- plain-C entry points of libeagle_host.so for Python (tests, bench.py): opaque pointers,
// integers and byte buffers only.  Two groups:
//   eh_backend_* / eh_ctx_* / eh_op_*   build and run single graphs on any ABI-compatible backend
//   eh_model_* / eh_spec_*              (model.cpp / driver.cpp) synthetic models and the speculative driver
#include "minihost.h"
#include <cstring>
#include <string>

#define EH_API extern "C" __attribute__((visibility("default")))
using namespace mh;

EH_API void * eh_backend_load(const char * path, const char * entry, int dev, char * err, int errlen) {
    std::string e;
    Backend * b = Backend::load(path, entry, dev, &e);
    if (!b && err && errlen > 0) snprintf(err, errlen, "%s", e.c_str());
    return b;
}
EH_API void eh_backend_free(void * b) { delete (Backend *) b; }
EH_API const char * eh_backend_name(void * b) { return ((Backend *) b)->name(); }
EH_API void eh_backend_set_threads(void * b, int n) { ((Backend *) b)->set_n_threads(n); }
EH_API int eh_backend_is_host(void * b) { return ((Backend *) b)->is_host; }
EH_API const char * eh_backend_description(void * b) { Backend * be = (Backend *) b; return be->dev->iface.get_description(be->dev); }
EH_API void eh_backend_memory(void * b, size_t * fr, size_t * tot) { Backend * be = (Backend *) b; be->dev->iface.get_memory(be->dev, fr, tot); }

EH_API int eh_ctx_use_split(void * c, int main_device, const float * tensor_split) { return ((Ctx *) c)->use_split(main_device, tensor_split) ? 1 : 0; }
EH_API void * eh_ctx_new(void * be, int usage) { Ctx * c = new Ctx((Backend *) be); c->usage = usage; return c; }
EH_API void eh_ctx_free(void * c) { delete (Ctx *) c; }
EH_API void * eh_tensor_new(void * c, int type, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) { return ((Ctx *) c)->new_tensor(type, ne0, ne1, ne2, ne3); }
EH_API void eh_tensor_set_name(void * c, void * t, const char * name) { ((Ctx *) c)->set_name((ggml_tensor *) t, name); }
EH_API void eh_tensor_set_flags(void * t, int flags) { ((ggml_tensor *) t)->flags = flags; }
EH_API void * eh_view(void * c, void * a, int nd, const int64_t * ne, const int64_t * nb, int64_t off) {
    size_t nbs[4] = {0, 0, 0, 0}; for (int i = 0; i < 4 && i < nd; ++i) nbs[i] = (size_t) nb[i];
    return ((Ctx *) c)->view((ggml_tensor *) a, nd, ne, nbs, (size_t) off);
}
EH_API void * eh_reshape(void * c, void * a, int64_t ne0, int64_t ne1, int64_t ne2, int64_t ne3) { return ((Ctx *) c)->reshape((ggml_tensor *) a, ne0, ne1, ne2, ne3); }
EH_API void * eh_permute(void * c, void * a, int a0, int a1, int a2, int a3) { return ((Ctx *) c)->permute((ggml_tensor *) a, a0, a1, a2, a3); }
EH_API void * eh_transpose(void * c, void * a) { return ((Ctx *) c)->transpose((ggml_tensor *) a); }
EH_API void * eh_cont(void * c, void * a) { return ((Ctx *) c)->cont((ggml_tensor *) a); }
EH_API void * eh_cpy(void * c, void * a, void * b) { return ((Ctx *) c)->cpy((ggml_tensor *) a, (ggml_tensor *) b); }
EH_API void * eh_mul_mat(void * c, void * a, void * b) { return ((Ctx *) c)->mul_mat((ggml_tensor *) a, (ggml_tensor *) b); }
EH_API void * eh_rms_norm(void * c, void * a, float eps) { return ((Ctx *) c)->rms_norm((ggml_tensor *) a, eps); }
EH_API void * eh_bin(void * c, int op, void * a, void * b) { return ((Ctx *) c)->bin(op, (ggml_tensor *) a, (ggml_tensor *) b); }
EH_API void * eh_unary(void * c, void * a, int uop) { return ((Ctx *) c)->unary((ggml_tensor *) a, uop); }
EH_API void * eh_scale(void * c, void * a, float s) { return ((Ctx *) c)->scale((ggml_tensor *) a, s); }
EH_API void * eh_concat(void * c, void * a, void * b, int dim) { return ((Ctx *) c)->concat((ggml_tensor *) a, (ggml_tensor *) b, dim); }
EH_API void * eh_argmax(void * c, void * a) { return ((Ctx *) c)->argmax((ggml_tensor *) a); }
EH_API void * eh_get_rows(void * c, void * a, void * b) { return ((Ctx *) c)->get_rows((ggml_tensor *) a, (ggml_tensor *) b); }
EH_API void * eh_rope(void * c, void * a, void * pos, void * ff, int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale,
                      float ext_factor, float attn_factor, float beta_fast, float beta_slow) {
    return ((Ctx *) c)->rope_ext((ggml_tensor *) a, (ggml_tensor *) pos, (ggml_tensor *) ff, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
}
EH_API void * eh_soft_max(void * c, void * a, void * mask, float scale, float max_bias) { return ((Ctx *) c)->soft_max_ext((ggml_tensor *) a, (ggml_tensor *) mask, scale, max_bias); }
// k best entries of rows of a device tensor through the backend's top-k extension (0 = done, -1 = the backend has none / declines)
EH_API int eh_top_k(void * c, void * t, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals) { return ((Ctx *) c)->be->top_k((ggml_tensor *) t, rows, n_rows, k, ids, vals) ? 0 : -1; }
EH_API int  eh_alloc(void * c) { return ((Ctx *) c)->alloc() ? 0 : -1; }
EH_API int  eh_compute(void * c) { return (int) ((Ctx *) c)->compute(); }
EH_API void eh_set(void * c, void * t, const void * data, int64_t off, int64_t size) { ((Ctx *) c)->set((ggml_tensor *) t, data, (size_t) off, (size_t) size); }
EH_API void eh_get(void * c, void * t, void * data, int64_t off, int64_t size) { ((Ctx *) c)->get((ggml_tensor *) t, data, (size_t) off, (size_t) size); }
EH_API int  eh_supports(void * c, void * t) { return ((Ctx *) c)->be->supports_op((ggml_tensor *) t) ? 1 : 0; }
EH_API int64_t eh_nbytes(void * t) { return (int64_t) nbytes((ggml_tensor *) t); }
EH_API void eh_shape(void * t, int64_t * ne, int64_t * nb) { ggml_tensor * x = (ggml_tensor *) t; for (int i = 0; i < 4; ++i) { ne[i] = x->ne[i]; nb[i] = (int64_t) x->nb[i]; } }
EH_API int  eh_type(void * t) { return ((ggml_tensor *) t)->type; }
EH_API int  eh_n_nodes(void * c) { return (int) ((Ctx *) c)->nodes.size(); }
// plugin node hooks (include/ggml_mi355x.h: ggml_backend_mi355x_set_node_hooks) for hosts / tests driving the C API from Python
EH_API int eh_backend_set_node_hooks(void * b, void ** nodes, int n, void (*fn)(void *, const ggml_tensor *, void *), void * user) {
    return ((mh::Backend *) b)->set_node_hooks((const ggml_tensor * const *) nodes, n, fn, user) ? 0 : -1;
}
EH_API void * eh_tensor_data(void * t) { return ((ggml_tensor *) t)->data; }

// shape / type / op / name / sources of a tensor handle
EH_API void eh_tensor_info(void * tp, int64_t * ne4, int64_t * nb4, int * type, int * op, int * flags, char * name, int cap) {
    const ggml_tensor * t = (const ggml_tensor *) tp;
    for (int i = 0; i < 4; ++i) { ne4[i] = t->ne[i]; nb4[i] = (int64_t) t->nb[i]; }
    *type = (int) t->type; *op = (int) t->op; *flags = t->flags;
    if (name && cap > 0) { strncpy(name, t->name, (size_t) cap - 1); name[cap - 1] = 0; }
}
EH_API void * eh_tensor_src(void * tp, int s) { return (s >= 0 && s < GGML_MAX_SRC) ? ((ggml_tensor *) tp)->src[s] : nullptr; }
EH_API void * eh_tensor_view_src(void * tp) { return ((ggml_tensor *) tp)->view_src; }
