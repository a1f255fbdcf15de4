// tp.cpp -- tensor-parallel plumbing of the host: one process per GPU, RCCL over xGMI.
// The reference has no collective library at all (its -sm row is a peer-memcpy star through the main GPU,
// R/ggml/src/ggml-cuda/ggml-cuda.cu:1359-1667); here every rank holds 1/N of each weight (heads / n_ff slices, in
// whole quantised blocks) and the layer needs two all-reduces of [n_embd, T] fp32 (model.cpp cut points).
// librccl is resolved with dlopen at run time so that the host library itself has no ROCm link dependency.
#include "model.h"
#include <dlfcn.h>
#include <cstdio>
#include <cstring>

#define EH_API extern "C" __attribute__((visibility("default")))

namespace {
typedef struct { char internal[128]; } ncclUniqueId;
typedef void * ncclComm_t;
typedef int (*ncclGetUniqueId_t)(ncclUniqueId *);
typedef int (*ncclCommInitRank_t)(ncclComm_t *, int, ncclUniqueId, int);
typedef int (*ncclAllReduce_t)(const void *, void *, size_t, int /*dtype*/, int /*op*/, ncclComm_t, void * /*stream*/);
typedef int (*ncclBroadcast_t)(const void *, void *, size_t, int, int, ncclComm_t, void *);
typedef int (*ncclCommDestroy_t)(ncclComm_t);
typedef int (*ncclCommCount_t)(const ncclComm_t, int *);
typedef const char * (*ncclGetErrorString_t)(int);
enum { ncclFloat32 = 7, ncclInt32 = 2, ncclSum = 0 };

struct Rccl {
    void * dl = nullptr;
    ncclGetUniqueId_t get_id = nullptr; ncclCommInitRank_t init_rank = nullptr; ncclAllReduce_t all_reduce = nullptr;
    ncclBroadcast_t bcast = nullptr; ncclCommDestroy_t destroy = nullptr; ncclGetErrorString_t errstr = nullptr; ncclCommCount_t count = nullptr;
    bool load() {
        if (dl) return true;
        for (const char * n : { "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so" }) { dl = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (dl) break; }
        if (!dl) return false;
        get_id = (ncclGetUniqueId_t) dlsym(dl, "ncclGetUniqueId"); init_rank = (ncclCommInitRank_t) dlsym(dl, "ncclCommInitRank");
        all_reduce = (ncclAllReduce_t) dlsym(dl, "ncclAllReduce"); bcast = (ncclBroadcast_t) dlsym(dl, "ncclBroadcast");
        destroy = (ncclCommDestroy_t) dlsym(dl, "ncclCommDestroy"); count = (ncclCommCount_t) dlsym(dl, "ncclCommCount"); errstr = (ncclGetErrorString_t) dlsym(dl, "ncclGetErrorString");
        return get_id && init_rank && all_reduce;
    }
} g_rccl;

struct TpComm { ncclComm_t comm = nullptr; void * stream = nullptr; int rank = 0, size = 1; };

void rccl_allreduce(void * user, void * data, int64_t n) {
    TpComm * c = (TpComm *) user;
    const int rc = g_rccl.all_reduce(data, data, (size_t) n, ncclFloat32, ncclSum, c->comm, c->stream);
    if (rc != 0) { fprintf(stderr, "[eagle_host] ncclAllReduce failed: %s\n", g_rccl.errstr ? g_rccl.errstr(rc) : "?"); abort(); }
}
} // namespace

// rank 0: returns the 128-byte unique id to be broadcast to the other ranks by the launcher (torch.distributed)
EH_API int eh_tp_unique_id(char * out128) {
    if (!g_rccl.load()) return -1;
    ncclUniqueId id; if (g_rccl.get_id(&id) != 0) return -2;
    memcpy(out128, id.internal, 128); return 0;
}
// every rank: create the communicator and bind the model's all-reduce to RCCL on the backend's own HIP stream
EH_API void * eh_tp_init(void * backend, const char * id128, int rank, int size) {
    if (!g_rccl.load()) return nullptr;
    mh::Backend * be = (mh::Backend *) backend;
    typedef void * (*stream_fn)(ggml_backend_t);
    stream_fn sf = be->reg->iface.get_proc_address ? (stream_fn) be->reg->iface.get_proc_address(be->reg, "ggml_backend_mi355x_stream") : nullptr;
    if (!sf) return nullptr;
    TpComm * c = new TpComm; c->rank = rank; c->size = size; c->stream = sf(be->be);
    ncclUniqueId id; memcpy(id.internal, id128, 128);
    if (g_rccl.init_rank(&c->comm, size, id, rank) != 0) { delete c; return nullptr; }
    return c;
}
EH_API void eh_tp_bind(void * comm, void * model) { eh::Model * m = (eh::Model *) model; m->allreduce = rccl_allreduce; m->allreduce_user = comm; }
// ranks in the communicator as RCCL itself reports them (bench.py prints it next to the number of processes it started)
EH_API int eh_tp_comm_size(void * comm) { TpComm * c = (TpComm *) comm; int n = 0; if (!c || !c->comm || !g_rccl.count || g_rccl.count(c->comm, &n) != 0) return -1; return n; }
EH_API void eh_tp_free(void * comm) { TpComm * c = (TpComm *) comm; if (c) { if (c->comm && g_rccl.destroy) g_rccl.destroy(c->comm); delete c; } }
