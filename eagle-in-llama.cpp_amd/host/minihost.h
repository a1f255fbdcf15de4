// minihost.h -- the host side above the C ABI: a small C++ counterpart of the parts of ggml that sit
// above the backend vtables (tensor/op constructors, graph assembly, buffer allocation, backend
// loading).  It talks to ANY library that implements the reference's backend ABI (include/ggml_abi.h):
// our MI355X plugin ("ggml_backend_init") and, in tests and for the CPU baseline, the reference's own
// CPU backend built under oracle/_ref ("ggml_backend_cpu_reg") -- the very same graphs run on both.
//
// Constructors fill ggml_tensor exactly as the reference's do (names, op_params slots, view rules):
//   ggml_new_tensor / view / reshape / permute / transpose / cont / cpy     R/ggml/src/ggml.c
//   ggml_mul_mat, ggml_rms_norm, ggml_rope_ext, ggml_soft_max_ext, ...      R/ggml/src/ggml.c
#pragma once
#include "ggml_abi.h"
#include <cstddef>
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <vector>

namespace mh {

struct type_traits { int blck; int size; };
type_traits traits(int type);
size_t row_size(int type, int64_t ne0);
size_t nbytes(const ggml_tensor * t);
int64_t nelements(const ggml_tensor * t);
bool   is_contiguous(const ggml_tensor * t);

// A loaded backend: registry -> device -> stream, plus its default buffer type.
struct Backend {
    void * dl = nullptr;
    ggml_backend_reg_t reg = nullptr;
    ggml_backend_dev_t dev = nullptr;
    ggml_backend_t     be  = nullptr;
    ggml_backend_buffer_type_t buft = nullptr;
    std::string path, entry;
    bool is_host = false;

    static Backend * load(const char * so_path, const char * entry_symbol, int device_index, std::string * err);
    ~Backend();
    const char * name() const;
    void set_n_threads(int n);                 // via get_proc_address("ggml_backend_set_n_threads") when offered
    // k best entries of rows of a device tensor through the plugin's "ggml_backend_mi355x_top_k" extension; false when the backend has
    // none (reference CPU backend) or declines the operands -- the caller then reads the rows back
    bool top_k(const ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals);
    // "ggml_backend_mi355x_set_node_hooks": fn(user, t, stream) from inside graph_compute after the node producing nodes[i]; false when
    // the backend has no such extension (reference CPU backend: the caller then computes the graph in segments)
    typedef void (*node_hook_fn)(void * user, const ggml_tensor * t, void * stream);
    bool set_node_hooks(const ggml_tensor * const * nodes, int n, node_hook_fn fn, void * user);
    void synchronize();
    bool supports_op(const ggml_tensor * t) const;
    ggml_backend_buffer_t alloc_buffer(size_t size, int usage);
    void free_buffer(ggml_backend_buffer_t b);
    // page-locked host memory from the device's host buffer type (what llama.cpp uses for its input / output staging,
    // R/src/llama.cpp:9210-9240); plain malloc when the backend has none (CPU).  Freed with host_free.
    void * host_alloc(size_t size);
    void   host_free(void * p);
  private:
    struct host_block { void * p; ggml_backend_buffer_t buf; };
    std::vector<host_block> host_blocks;
};

// grow-only float array in page-locked memory: async device->host copies land here
struct HostVec {
    Backend * be = nullptr; float * p = nullptr; size_t n = 0, cap = 0;
    ~HostVec() { if (p && be) be->host_free(p); }
    void resize(size_t m) { if (m > cap) { if (p) { be->synchronize(); be->host_free(p); } cap = m + m/2 + 1024; p = (float *) be->host_alloc(cap * sizeof(float)); } n = m; }
    void clear() { n = 0; }
    float * data() { return p; } const float * data() const { return p; }
    float * begin() { return p; } const float * begin() const { return p; }
    size_t size() const { return n; }
    float & operator[](size_t i) { return p[i]; } const float & operator[](size_t i) const { return p[i]; }
};

// Tensor arena + op list + device memory for one graph (or for a set of persistent tensors).
struct Ctx {
    Backend * be;
    // tensor arena: chunks of 256 structs that survive reset_graph() (a per-step graph is ~1000 tensors: no malloc per tensor, no
    // re-faulting of the pages on every decode); pointers stay valid until the Ctx dies
    struct TensorPool {
        static constexpr size_t CHUNK = 256;
        std::vector<std::unique_ptr<ggml_tensor[]>> chunks; size_t n = 0;
        ggml_tensor * push() { if (n == chunks.size() * CHUNK) chunks.emplace_back(new ggml_tensor[CHUNK]); ggml_tensor * t = &chunks[n / CHUNK][n % CHUNK]; ++n; return t; }
        void clear() { n = 0; }
        size_t size() const { return n; }
        ggml_tensor & operator[](size_t i) { return chunks[i / CHUNK][i % CHUNK]; }
        struct iterator { TensorPool * p; size_t i; ggml_tensor & operator*() { return (*p)[i]; } iterator & operator++() { ++i; return *this; } bool operator!=(const iterator & o) const { return i != o.i; } };
        iterator begin() { return { this, 0 }; } iterator end() { return { this, n }; }
    } pool;
    std::vector<ggml_tensor *> nodes;            // ops in creation order == execution order
    std::vector<ggml_backend_buffer_t> buffers;  // owned
    std::vector<ggml_tensor **> node_ptrs_storage;
    int usage = GGML_BACKEND_BUFFER_USAGE_ANY;
    ggml_backend_buffer_type_t buft_override = nullptr;      // e.g. the row-split buffer type (-sm row): tensors of this Ctx are allocated there
    // R/src/llama-model.cpp:310-322: ask the registry for "ggml_backend_split_buffer_type" and use what it returns for the weights
    bool use_split(int main_device, const float * tensor_split);

    explicit Ctx(Backend * b) : be(b) {}
    ~Ctx();

    ggml_tensor * new_tensor(int type, int64_t ne0, int64_t ne1 = 1, int64_t ne2 = 1, int64_t ne3 = 1, const char * name = nullptr);
    // views (no data of their own)
    ggml_tensor * view(ggml_tensor * a, int n_dims, const int64_t * ne, const size_t * nb /* nb[1..n_dims-1] */, size_t offset);
    ggml_tensor * view_1d(ggml_tensor * a, int64_t ne0, size_t offset);
    ggml_tensor * view_2d(ggml_tensor * a, int64_t ne0, int64_t ne1, size_t nb1, size_t offset);
    ggml_tensor * view_3d(ggml_tensor * a, int64_t ne0, int64_t ne1, int64_t ne2, size_t nb1, size_t nb2, size_t offset);
    ggml_tensor * reshape(ggml_tensor * a, int64_t ne0, int64_t ne1 = 1, int64_t ne2 = 1, int64_t ne3 = 1);
    ggml_tensor * permute(ggml_tensor * a, int ax0, int ax1, int ax2, int ax3);
    ggml_tensor * transpose(ggml_tensor * a);
    // ops
    ggml_tensor * cont(ggml_tensor * a);
    ggml_tensor * cont_2d(ggml_tensor * a, int64_t ne0, int64_t ne1);
    ggml_tensor * cpy(ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * mul_mat(ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * rms_norm(ggml_tensor * a, float eps);
    ggml_tensor * add(ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * mul(ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * bin(int op, ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * unary(ggml_tensor * a, int uop);
    ggml_tensor * scale(ggml_tensor * a, float s);
    ggml_tensor * concat(ggml_tensor * a, ggml_tensor * b, int dim);
    ggml_tensor * get_rows(ggml_tensor * a, ggml_tensor * b);
    ggml_tensor * argmax(ggml_tensor * a);        // ggml_argmax: i32 [ne1] indices of the row maxima of a 2-D tensor
    ggml_tensor * rope_ext(ggml_tensor * a, ggml_tensor * pos, ggml_tensor * ff, int n_dims, int mode, int n_ctx_orig,
                           float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow);
    ggml_tensor * soft_max_ext(ggml_tensor * a, ggml_tensor * mask, float scale, float max_bias);

    void set_name(ggml_tensor * t, const char * name);
    // give every tensor that owns data and has none yet a place in one new backend buffer
    bool alloc();
    // drop all ops/tensors but keep the buffers for re-use by the next alloc() (per-step graphs)
    void reset_graph();
    void set(ggml_tensor * t, const void * data, size_t offset, size_t size);
    void get(const ggml_tensor * t, void * data, size_t offset, size_t size);
    // ordered on the backend's queue (ggml_backend_tensor_{set,get}_async); synchronous fall-back when the backend has no async copies.
    // `data` must stay valid until the next synchronize().
    void set_async(ggml_tensor * t, const void * data, size_t offset, size_t size);
    void get_async(const ggml_tensor * t, void * data, size_t offset, size_t size);
    enum ggml_status compute();                  // graph_compute over `nodes` + synchronize
    enum ggml_status compute_async();
    enum ggml_status compute_range(int n0, int n1);   // nodes [n0, n1) only, asynchronous (tensor-parallel segments)
    double t_issue_us = 0, t_wait_us = 0;        // host time inside graph_compute vs waiting for the device

  private:
    ggml_tensor * op_result(int op, int type, const int64_t * ne, ggml_tensor * a, ggml_tensor * b = nullptr, ggml_tensor * c = nullptr);
    ggml_tensor * view_impl(ggml_tensor * a, int op);
    size_t reuse_cap = 0;                        // capacity of buffers.back() when it is the reusable compute buffer
    bool   reuse = false;
    std::vector<ggml_tensor *> node_array;
    ggml_cgraph graph{};
};

} // namespace mh
