// mi355x_common.h -- shared host-side helpers of the MI355X ggml backend plugin.
#pragma once
#include "ggml_abi.h"
#include <hip/hip_runtime_api.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define MI_LOG(fmt, ...)  fprintf(stderr, "[ggml-mi355x] " fmt "\n", ##__VA_ARGS__)
#define MI_ABORT(fmt, ...) do { MI_LOG("FATAL %s:%d: " fmt, __FILE__, __LINE__, ##__VA_ARGS__); abort(); } while (0)
#define MI_ASSERT(x) do { if (!(x)) MI_ABORT("assertion failed: %s", #x); } while (0)
// Same convention as the reference GPU backend: a runtime error is fatal
// (R/ggml/src/ggml-cuda/ggml-cuda.cu:65 ggml_cuda_error -> abort).
#define HIP_CHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
    MI_ABORT("%s -> %s", #call, hipGetErrorString(e_)); } while (0)

// ---- element-type traits (R/ggml/src/ggml.c type_traits[]: blck_size / type_size) ----
struct mi_type_traits { int blck; int size; const char * name; };
static inline mi_type_traits mi_traits(int type) {
    switch (type) {
        case GGML_TYPE_F32:  return {1, 4, "f32"};
        case GGML_TYPE_F16:  return {1, 2, "f16"};
        case GGML_TYPE_BF16: return {1, 2, "bf16"};
        case GGML_TYPE_I8:   return {1, 1, "i8"};
        case GGML_TYPE_I16:  return {1, 2, "i16"};
        case GGML_TYPE_I32:  return {1, 4, "i32"};
        case GGML_TYPE_I64:  return {1, 8, "i64"};
        case GGML_TYPE_F64:  return {1, 8, "f64"};
        case GGML_TYPE_Q4_0: return {32, 18, "q4_0"};
        case GGML_TYPE_Q4_1: return {32, 20, "q4_1"};
        case GGML_TYPE_Q5_0: return {32, 22, "q5_0"};
        case GGML_TYPE_Q5_1: return {32, 24, "q5_1"};
        case GGML_TYPE_Q8_0: return {32, 34, "q8_0"};
        case GGML_TYPE_Q8_1: return {32, 36, "q8_1"};
        case GGML_TYPE_Q2_K: return {256, 84, "q2_K"};
        case GGML_TYPE_Q3_K: return {256, 110, "q3_K"};
        case GGML_TYPE_Q4_K: return {256, 144, "q4_K"};
        case GGML_TYPE_Q5_K: return {256, 176, "q5_K"};
        case GGML_TYPE_Q6_K: return {256, 210, "q6_K"};
        case GGML_TYPE_Q8_K: return {256, 292, "q8_K"};
        default:             return {0, 0, "?"};
    }
}
static inline bool   mi_is_quantized(int type) { return mi_traits(type).blck > 1; }
static inline size_t mi_row_size(int type, int64_t ne) { auto t = mi_traits(type); return (size_t)(ne / t.blck) * t.size; }
static inline int64_t mi_nelements(const ggml_tensor * t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }
static inline int64_t mi_nrows(const ggml_tensor * t) { return t->ne[1]*t->ne[2]*t->ne[3]; }
// ggml_nbytes (R/ggml/src/ggml.c "size_t ggml_nbytes"): span of the tensor incl. strides
static inline size_t mi_nbytes(const ggml_tensor * t) {
    auto tr = mi_traits(t->type);
    if (tr.blck == 0) return 0;
    for (int i = 0; i < GGML_MAX_DIMS; ++i) if (t->ne[i] <= 0) return 0;
    size_t n;
    if (tr.blck == 1) {
        n = tr.size;
        for (int i = 0; i < GGML_MAX_DIMS; ++i) n += (t->ne[i] - 1) * t->nb[i];
    } else {
        n = t->ne[0] * t->nb[0] / tr.blck;
        for (int i = 1; i < GGML_MAX_DIMS; ++i) n += (t->ne[i] - 1) * t->nb[i];
    }
    return n;
}
static inline bool mi_is_contiguous(const ggml_tensor * t) {
    auto tr = mi_traits(t->type);
    if (tr.blck == 0) return false;
    size_t next = tr.size;
    if (t->ne[0] != tr.blck && t->nb[0] != next) return false;
    next *= t->ne[0] / tr.blck;
    for (int i = 1; i < GGML_MAX_DIMS; ++i) {
        if (t->ne[i] != 1) { if (t->nb[i] != next) return false; next *= t->ne[i]; }
        else next = t->ne[i] * next;
    }
    return true;
}
static inline bool mi_same_shape(const ggml_tensor * a, const ggml_tensor * b) {
    return a->ne[0]==b->ne[0] && a->ne[1]==b->ne[1] && a->ne[2]==b->ne[2] && a->ne[3]==b->ne[3];
}
static inline float   mi_op_f32(const ggml_tensor * t, int i) { float v; memcpy(&v, &t->op_params[i], 4); return v; }
static inline int32_t mi_op_i32(const ggml_tensor * t, int i) { return t->op_params[i]; }

#define MI_MAX_DEVICES 16
// Lab switches.  The shipped plugin is built WITHOUT -DMI_LAB: mi_lab_env() is then a constant nullptr, the A/B knobs read through it fold to
// their defaults and the diagnostic kernel instantiations (phase stamps) are not in the code object.  `python eagle-in-llama.cpp_amd/build.py --lab`
// builds the lab variant (lib/libggml-mi355x-lab.so) that scripts/ab_bench.py, mmt_stamps.py, attn_stamps.py ... load.
// Product settings stay plain getenv: GGML_MI355X_NO_FUSION, _NO_TILE, _SYNC_SPIN, _SPLIT_FAKE_DEVICES, _SPLIT_STAGE.
#ifdef MI_LAB
static inline const char * mi_lab_env(const char * name) { return getenv(name); }
#else
static inline const char * mi_lab_env(const char *) { return nullptr; }
#endif
// ---- plugin-wide objects (defined in backend.cpp) ----
struct mi_device_ctx {            // ggml_backend_device::context
    int  device;                  // HIP ordinal
    char name[32];                // "MI355X<n>"
    char desc[256];
};
struct mi_backend_ctx {           // ggml_backend::context  (one HIP stream)
    int          device;
    hipStream_t  stream;
    void *       scratch;         // device scratch (activation quantisation etc.)
    size_t       scratch_size;
    struct mi_act_cache * act_cache;   // kernels.h
    char         name[32];
    hipEvent_t   copy_ev;         // cross-backend copies: recorded on this stream, waited for by the destination's (created at first use)
    // node hooks ("ggml_backend_mi355x_set_node_hooks", include/ggml_mi355x.h): graph_compute calls node_hook(user, t, stream) right after the
    // work of the node producing one of hook_nodes[] has been queued -- a host enqueues its RCCL all-reduce there, a forward stays ONE submission
    void       (*node_hook)(void * user, const ggml_tensor * t, void * stream);
    void *       node_hook_user;
    const ggml_tensor * const * hook_nodes;
    int          n_hook_nodes;
    struct mi_split_events * split_ev;    // split.cpp: join events of row-split MUL_MATs issued through this backend (created at first use)
};
struct mi_buffer_ctx {            // ggml_backend_buffer::context
    int    device;
    void * base;
    size_t size;
    bool   host;                  // pinned host buffer
};

bool mi_buffer_is_ours(ggml_backend_buffer_t buf);          // device buffer of this plugin
bool mi_buft_is_ours(ggml_backend_buffer_type_t buft);
bool mi_buft_is_our_host(ggml_backend_buffer_type_t buft);
void * mi_scratch(mi_backend_ctx * ctx, size_t size);       // grow-only scratch on ctx->device

// row-split weight buffers (split.cpp)
ggml_backend_buffer_t mi_make_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * ctx, size_t size);
int  mi_device_count();
ggml_backend_dev_t mi_device(int i);                        // i = index among the kept (gfx950) devices
int  mi_device_ordinal(int i);                              // its HIP ordinal (-1: no such device)
bool mi_buft_is_split(ggml_backend_buffer_type_t buft);
bool mi_tensor_is_split(const ggml_tensor * t);
bool mi_split_supports_mul_mat(const ggml_tensor * op);
void mi_split_free_events(mi_backend_ctx * ctx);
void mi_split_mul_mat(mi_backend_ctx * ctx, const ggml_tensor * dst);          // dst = MUL_MAT(split weight, f32 activations), gathered on ctx's device

// weight re-layout (backend.cpp / kernels_tile.hip / tile_layout.h)
bool mi_is_tiled(const ggml_tensor * t);                    // t itself carries the tiled tag
bool mi_tile_eligible(const ggml_tensor * w);
bool mi_ensure_tiled(ggml_tensor * w);                      // first MUL_MAT use: permute in place; returns whether w is tiled now
void mi_tile_release_scratch(int dev);                      // end of a graph: give the conversion scratch back
void mi_untile(ggml_tensor * t, bool never_again);         // back to ggml's layout (t or the tensor it views)
// hipFuncAttributeMaxDynamicSharedMemorySize for `fn` on the CURRENT device, once per (kernel, device): the attribute lives in the
// device's code object, so a kernel first launched on device 1 (row-split buffers) needs its own call
void mi_allow_big_lds(const void * fn);
void mi_tile_permute(hipStream_t st, const void * src, void * dst, int type, int64_t rows, int64_t k, bool fwd);

// graph.cpp
enum ggml_status mi_graph_compute(mi_backend_ctx * ctx, ggml_cgraph * g);
bool             mi_supports_op(int device, const ggml_tensor * op);
