/*
 * ggml_mi355x.h -- exported C ABI of libggml-mi355x.so, the MI355X (gfx950) backend plugin that makes
 * llama_decode() / llama_decode_draft() / the speculative EAGLE drivers of mkjsym/EAGLE-in-llama.cpp run
 * on AMD Instinct MI355X without touching the reference's sources.
 *
 * Every entry point below is one the reference's own loader binds; nothing else is needed to drop the
 * library in (R = /root/reference/llama.cpp):
 *
 *   ggml_backend_init     required by  R/ggml/src/ggml-backend-reg.cpp:237  (dlsym "ggml_backend_init"),
 *                         contract in  R/ggml/src/ggml-backend-impl.h:215,220-228 (GGML_BACKEND_DL_IMPL).
 *                         Returns the registry object; reg->api_version must be 1 (reg.cpp:246).
 *   ggml_backend_score    optional,    R/ggml/src/ggml-backend-reg.cpp:229-235; 0 => "not usable here".
 *
 * Loading:  GGML_BACKEND_PATH=/path/libggml-mi355x.so <any reference binary>      (reg.cpp:578-581)
 *      or:  ggml_backend_load("/path/libggml-mi355x.so")                          (reg.cpp:393)
 *      or:  copy next to the executable as libggml-hip-mi355x.so                  (reg.cpp:568 name probe)
 *
 * Everything else crosses the boundary through the five vtables reachable from the returned registry
 * (layout stated in ggml_abi.h): get_device -> {init_backend, get_buffer_type, supports_op, ...} ->
 * ggml_backend_i::graph_compute, which is the hot path (R/ggml/src/ggml-backend.cpp:1397).
 *
 * Named extension points the reference looks up with reg->iface.get_proc_address():
 *   "ggml_backend_split_buffer_type"  (R/src/llama-model.cpp:310-322, -sm row)  -> ggml_backend_mi355x_split_buffer_type
 *   "ggml_backend_get_features"       (R/src/llama.cpp:12044)                    -> feature list
 * and one extension of ours, "ggml_backend_mi355x_stream": void * (*)(ggml_backend_t) -> the hipStream_t of a backend
 * instance, for hosts that enqueue RCCL collectives between graph segments (tensor parallel, host/tp.cpp);
 * "ggml_backend_mi355x_set_node_hooks" (below) so that those collectives are enqueued from inside graph_compute,
 * and "ggml_backend_mi355x_top_k" (below) for hosts that draft token trees.
 */
#ifndef GGML_MI355X_H
#define GGML_MI355X_H

#include "ggml_abi.h"

#ifdef __cplusplus
extern "C" {
#endif

#define GGML_MI355X_API __attribute__((visibility("default")))

/* dlopen protocol of the reference (see above) */
GGML_MI355X_API ggml_backend_reg_t ggml_backend_init(void);
GGML_MI355X_API int                ggml_backend_score(void);

/* static-link style accessors, same role as ggml_backend_cuda_reg() / ggml_backend_cuda_get_device_count()
 * (R/ggml/include/ggml-cuda.h:24-47) for hosts that link the library instead of dlopen-ing it */
GGML_MI355X_API ggml_backend_reg_t ggml_backend_mi355x_reg(void);
GGML_MI355X_API int                ggml_backend_mi355x_device_count(void);

/* row-split (tensor-parallel) weight buffers, signature of ggml_backend_split_buffer_type_t
 * (R/ggml/include/ggml-backend.h:188); tensor_split has one entry per device, NULL => even split */
GGML_MI355X_API ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split);

/* Device-side top-k for tree drafting: the reference's tree driver reads the k best candidates of every live branch from the sampler's
 * sorted cur_p (R/common/speculative.cpp:257-272, R/examples/speculative/speculative-eagle.cpp:542-625) after llama_get_logits_ith has
 * brought the whole row (n_vocab floats) to the host.  `logits`: f32 tensor [n_vocab <= 65536, n_outputs] in a device buffer of this plugin;
 * rows: n_rows row indices (NULL: 0 .. n_rows-1); for every row the k <= 64 largest values in descending order, ties: the lower index first;
 * ids / vals: HOST arrays [n_rows][k] (fewer than k entries in a row: id -1, value -inf).  Ordered behind everything already submitted to
 * `backend`, returns when the results are on the host.  0 = ok, -1 = not supported for these operands (the caller keeps its host path). */
GGML_MI355X_API int ggml_backend_mi355x_top_k(ggml_backend_t backend, const struct ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals);

/* Extension "ggml_backend_mi355x_set_node_hooks": collectives from INSIDE graph_compute.  `fn(user, t, stream)` is called when the node that
 * produces nodes[i] has been queued on the backend's HIP stream; whatever fn enqueues on `stream` (ncclAllReduce in place on t->data)
 * is ordered before every later node of the graph, so a tensor-parallel forward is one graph_compute instead of one per all-reduce
 * (host/tp.cpp, host/model.cpp).  The reference has no counterpart: its -sm row gathers by peer copies inside ggml_cuda_op_mul_mat
 * (R/ggml/src/ggml-cuda/ggml-cuda.cu:1590-1666).  Mark the tensors GGML_TENSOR_FLAG_OUTPUT so that no fusion swallows them;
 * nodes == NULL removes the hooks; the array must outlive them.  Returns 0, or -1 for a backend of another plugin. */
GGML_MI355X_API int ggml_backend_mi355x_set_node_hooks(ggml_backend_t backend, const struct ggml_tensor * const * nodes, int n,
                                                       void (*fn)(void * user, const struct ggml_tensor * t, void * stream), void * user);

/* measurement hook (not part of the reference's interface): HIP-event timing of every quantised mat-vec launch
 * between begin/end, on the stream the kernels are launched on.  out[0] = kernel milliseconds, out[1] = algorithmic
 * bytes (weights + fp32 activations + outputs); returns the number of launches.  Used by bench.py. */
GGML_MI355X_API void ggml_backend_mi355x_profile_begin(void);
GGML_MI355X_API int  ggml_backend_mi355x_profile_end(double * out);
/* profile_end with the launches split by batch size: out[0..3] = {kernel ms, algorithmic bytes, int8 ops (2 T rows k), launches} of the
 * launches with fewer than min_tokens tokens, out[4..7] the same of the others, out[8] = ms of an empty event pair.  bench.py prices the
 * big-batch GEMM of a tree verification with it (min_tokens = 25). */
GGML_MI355X_API int  ggml_backend_mi355x_profile_end_by_batch(int min_tokens, double * out);
/* count-only variant for runs under rocprofv3: no events are inserted (the launch sequence is the product's); between begin and end the
 * plugin counts quantised mat-vec launches and their algorithmic bytes, and both calls launch the marker kernel `k_profile_mark`, so the
 * kernel trace of the same process can be cut at exactly these points (bench.py --rocprof-child). */
GGML_MI355X_API void ggml_backend_mi355x_count_begin(void);
GGML_MI355X_API long ggml_backend_mi355x_count_end(double * bytes);
/* row-split arithmetic of the split buffer type (get_row_split, R/ggml/src/ggml-cuda/ggml-cuda.cu:735-748), for host-side tests */
GGML_MI355X_API void ggml_backend_mi355x_row_split(int64_t nrows, const float * tensor_split, int n_dev, int id, int64_t * lo, int64_t * hi);

#ifdef __cplusplus
}
#endif
#endif
