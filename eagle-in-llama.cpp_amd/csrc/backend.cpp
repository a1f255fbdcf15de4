// backend.cpp -- the drop-in boundary: the five ggml backend vtables for MI355X (gfx950).
//
// Mirrors the INTERFACE of the reference GPU backend, not its code:
//   registry        R/ggml/src/ggml-cuda/ggml-cuda.cu:3383-3470   (get_proc_address names, static reg)
//   device          :2868-3300   (props: async/host_buffer/events, supports_op, offload_op >= 32 tokens)
//   buffer type     :620-715     (alignment 128, alloc returns NULL on OOM)
//   buffer          :520-618     (set/get synchronous on a per-thread stream, cpy_tensor D2D)
//   host buffer     :1050-1124   (pinned memory; is_host)
//   backend/stream  :2322-2412   (async set/get, synchronize), events :2769-2792, :3258-3286
//   dlopen protocol R/ggml/src/ggml-backend-impl.h:215-247 and R/ggml/src/ggml-backend-reg.cpp:229-247
#include "mi355x_common.h"
#include "ggml_mi355x.h"
#include "kernels.h"
#include <mutex>
#include <chrono>
#include <vector>
#include <string>
#include <dlfcn.h>
#include "tile_layout.h"


// The host's own constructor for buffer objects (R/ggml/src/ggml-backend.cpp "ggml_backend_buffer_init").
// Resolved from the process that loaded us when it is a ggml host, so that the object is created and
// later deleted by the same library; absent (stand-alone hosts) we build the object ourselves.
typedef ggml_backend_buffer_t (*host_buffer_init_t)(ggml_backend_buffer_type_t, struct ggml_backend_buffer_i, void *, size_t);
static ggml_backend_buffer_t make_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * ctx, size_t size) {
    static host_buffer_init_t host_init = (host_buffer_init_t) dlsym(RTLD_DEFAULT, "ggml_backend_buffer_init");
    if (host_init) return host_init(buft, iface, ctx, size);
    ggml_backend_buffer * b = new ggml_backend_buffer;
    b->iface = iface; b->buft = buft; b->context = ctx; b->size = size; b->usage = GGML_BACKEND_BUFFER_USAGE_ANY;
    return b;
}

static int g_ndev = -1;
static ggml_backend_device       g_devs[MI_MAX_DEVICES];
static mi_device_ctx             g_devctx[MI_MAX_DEVICES];
static ggml_backend_buffer_type  g_bufts[MI_MAX_DEVICES];
static ggml_backend_buffer_type  g_host_buft;
static ggml_backend_reg          g_reg;
static ggml_guid                 g_guid = { 0x4d, 0x49, 0x33, 0x35, 0x35, 0x58, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x2d, 0x30, 0x31 };

static void set_device(int dev) { HIP_CHECK(hipSetDevice(dev)); }
// for split.cpp (row-split buffers live outside the per-device buffer types)
ggml_backend_buffer_t mi_make_buffer(ggml_backend_buffer_type_t buft, const ggml_backend_buffer_i & iface, void * ctx, size_t size) { return make_buffer(buft, iface, ctx, size); }
int mi_device_count() { return g_ndev; }
ggml_backend_dev_t mi_device(int i) { return (i >= 0 && i < g_ndev) ? &g_devs[i] : nullptr; }
int mi_device_ordinal(int i) { return (i >= 0 && i < g_ndev) ? g_devctx[i].device : -1; }

// ============================================================ weight re-layout state (tile_layout.h, kernels_tile.hip)
// A quantised weight matrix keeps ggml's row-major blocks until the first MUL_MAT that reads it; mi_ensure_tiled() then permutes
// it once, in place, into 16-row x 1-unit tiles and tags the tensor through ggml_tensor::extra (which belongs to the backend
// that owns the buffer, as in the reference: ggml-cuda.cu:840).  Every other way to see the bytes goes through mi_untile() or
// an on-the-fly inverse, so hosts keep observing ggml's layout: get_tensor, cpy_tensor, partial set_tensor / memset.
// Threads: raw -> tiled is race-free -- mi_ensure_tiled() takes g_tile_mu, the conversion is synchronous (device drained before, stream
// drained after) and the tag flips last, so every launch path (all of them call mi_ensure_tiled first) sees either "raw, ineligible
// for good" or "tiled, conversion complete".  tiled -> raw happens only when a host writes part of the tensor, or a view of it becomes a
// mat-mul operand: touching a weight while another thread's graph reads it is a race under ggml's contract on any backend.
// tiled_read() (get_tensor) converts into scratch and leaves the tensor alone.  The scratch (as large as the largest weight converted)
// is given back at the end of the graph that used it (mi_tile_release_scratch): conversions are first-use events.
static char g_tag_tiled, g_tag_notile;
static std::mutex g_tile_mu;
static void * g_tile_scratch[MI_MAX_DEVICES] = {};
static size_t g_tile_scratch_size[MI_MAX_DEVICES] = {};
static bool tile_disabled() { static const bool v = getenv("GGML_MI355X_NO_TILE") != nullptr; return v; }
static void * tile_scratch(int dev, size_t n) {
    if (n > g_tile_scratch_size[dev]) {
        if (g_tile_scratch[dev]) HIP_CHECK(hipFree(g_tile_scratch[dev]));
        HIP_CHECK(hipMalloc(&g_tile_scratch[dev], n)); g_tile_scratch_size[dev] = n;
    }
    return g_tile_scratch[dev];
}
void mi_tile_release_scratch(int dev) {
    if (!g_tile_scratch[dev]) return;                                  // (unlocked peek: the common case, nothing was converted)
    std::lock_guard<std::mutex> lk(g_tile_mu);
    if (g_tile_scratch[dev]) { set_device(dev); HIP_CHECK(hipFree(g_tile_scratch[dev])); g_tile_scratch[dev] = nullptr; g_tile_scratch_size[dev] = 0; }
}
static inline ggml_tensor * tile_root(const ggml_tensor * t) { return (ggml_tensor *)(t->view_src ? t->view_src : t); }
bool mi_is_tiled(const ggml_tensor * t) { return t->extra == (void *) &g_tag_tiled; }
bool mi_tile_eligible(const ggml_tensor * w) {
    if (tile_disabled() || !w || w->view_src || w->op != GGML_OP_NONE || !w->data || !mi_buffer_is_ours(w->buffer)) return false;
    if (w->extra == (void *) &g_tag_notile) return false;
    if (w->buffer->usage == GGML_BACKEND_BUFFER_USAGE_COMPUTE) return false;        // scheduler copies of host-resident weights are rewritten every graph
    const int ub = mi_unit_bytes(w->type);
    if (!ub || w->ne[2] != 1 || w->ne[3] != 1 || (w->ne[1] % 16) || (w->ne[0] % 256) || w->ne[1] < 16) return false;
    return w->nb[0] == (size_t) mi_traits(w->type).size && w->nb[1] == mi_row_size(w->type, w->ne[0]);
}
// in place: raw -> tiled (fwd) or tiled -> raw; everything on the device is drained first (first use / rare paths only)
static void tile_convert(ggml_tensor * w, bool fwd) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) w->buffer->context; set_device(c->device);
    const size_t n = mi_nbytes(w);
    HIP_CHECK(hipDeviceSynchronize());
    void * tmp = tile_scratch(c->device, n);
    HIP_CHECK(hipMemcpyAsync(tmp, w->data, n, hipMemcpyDeviceToDevice, hipStreamPerThread));
    mi_tile_permute(hipStreamPerThread, tmp, w->data, w->type, w->ne[1], w->ne[0], fwd);
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
bool mi_ensure_tiled(ggml_tensor * w) {
    if (mi_is_tiled(w)) return true;
    std::lock_guard<std::mutex> lk(g_tile_mu);
    if (mi_is_tiled(w)) return true;
    if (!mi_tile_eligible(w)) return false;
    tile_convert(w, true);
    w->extra = (void *) &g_tag_tiled;
    return true;
}
// back to ggml's layout; `never_again` pins the tensor there (a view of it is used as a mat-mul operand)
void mi_untile(ggml_tensor * t, bool never_again) {
    ggml_tensor * w = tile_root(t);
    std::lock_guard<std::mutex> lk(g_tile_mu);
    if (mi_is_tiled(w)) { tile_convert(w, false); w->extra = nullptr; }
    if (never_again) w->extra = (void *) &g_tag_notile;
}
// the whole tensor is about to be overwritten: no need to convert what is there
static void tile_forget(ggml_tensor * w) { std::lock_guard<std::mutex> lk(g_tile_mu); if (mi_is_tiled(w)) w->extra = nullptr; }
static void before_write(ggml_tensor * t, size_t off, size_t size) {
    ggml_tensor * w = tile_root(t);
    if (!mi_is_tiled(w)) return;
    if (t == w && off == 0 && size == mi_nbytes(w)) tile_forget(w); else mi_untile(w, false);
}
// ggml-layout bytes [off, off+size) of a tiled tensor (or of a view into one) -> host
static void tiled_read(const ggml_tensor * t, void * data, size_t off, size_t size) {
    ggml_tensor * w = tile_root(t);
    std::lock_guard<std::mutex> lk(g_tile_mu);
    mi_buffer_ctx * c = (mi_buffer_ctx *) w->buffer->context; set_device(c->device);
    HIP_CHECK(hipDeviceSynchronize());
    char * tmp = (char *) tile_scratch(c->device, mi_nbytes(w));
    mi_tile_permute(hipStreamPerThread, w->data, tmp, w->type, w->ne[1], w->ne[0], false);
    HIP_CHECK(hipMemcpyAsync(data, tmp + ((const char *) t->data - (const char *) w->data) + off, size, hipMemcpyDeviceToHost, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}

// ============================================================ device buffer
static void buf_free(ggml_backend_buffer_t b) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) b->context;
    if (c->host) { HIP_CHECK(hipHostFree(c->base)); }
    else { set_device(c->device); HIP_CHECK(hipFree(c->base)); }
    delete c;
}
static void * buf_get_base(ggml_backend_buffer_t b) { return ((mi_buffer_ctx *) b->context)->base; }
static void buf_init_tensor(ggml_backend_buffer_t, ggml_tensor * t) {
    // (re)allocation: whatever lived at this address before is gone, and with it any layout tag of ours
    if (!t->view_src) t->extra = nullptr;
}
static void buf_memset_tensor(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t v, size_t off, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) b->context; set_device(c->device);
    if (!(off == 0 && size == mi_nbytes(t))) before_write(t, off, size);           // a constant fill of the whole tensor is layout-invariant
    HIP_CHECK(hipMemsetAsync((char *) t->data + off, v, size, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static void buf_set_tensor(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t off, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) b->context; set_device(c->device);
    before_write(t, off, size);
    HIP_CHECK(hipMemcpyAsync((char *) t->data + off, data, size, hipMemcpyHostToDevice, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static void buf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t off, size_t size) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) b->context; set_device(c->device);
    if (mi_is_tiled(tile_root(t))) { tiled_read(t, data, off, size); return; }
    HIP_CHECK(hipMemcpyAsync(data, (const char *) t->data + off, size, hipMemcpyDeviceToHost, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
}
static bool buf_cpy_tensor(ggml_backend_buffer_t b, const ggml_tensor * src, ggml_tensor * dst) {
    if (!mi_buffer_is_ours(src->buffer)) return false;                 // caller falls back to get+set
    if (mi_is_tiled(tile_root(src))) return false;                     // get_tensor hands out ggml's layout
    before_write(dst, 0, mi_nbytes(dst));
    mi_buffer_ctx * sc = (mi_buffer_ctx *) src->buffer->context, * dc = (mi_buffer_ctx *) b->context;
    const size_t n = mi_nbytes(src);
    set_device(dc->device);
    if (sc->device == dc->device) HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, hipStreamPerThread));
    else                          HIP_CHECK(hipMemcpyPeerAsync(dst->data, dc->device, src->data, sc->device, n, hipStreamPerThread));
    HIP_CHECK(hipStreamSynchronize(hipStreamPerThread));
    return true;
}
static void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    mi_buffer_ctx * c = (mi_buffer_ctx *) b->context; set_device(c->device);
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemset(c->base, v, c->size));
    HIP_CHECK(hipDeviceSynchronize());
}
static const ggml_backend_buffer_i g_buf_iface = {
    buf_free, buf_get_base, buf_init_tensor, buf_memset_tensor, buf_set_tensor, buf_get_tensor, buf_cpy_tensor, buf_clear, nullptr
};
bool mi_buffer_is_ours(ggml_backend_buffer_t buf) { return buf && buf->iface.get_base == buf_get_base && !((mi_buffer_ctx *) buf->context)->host; }

// ============================================================ device buffer type
static const char * buft_name(ggml_backend_buffer_type_t t) { return ((mi_device_ctx *) t->device->context)->name; }
static ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    mi_device_ctx * d = (mi_device_ctx *) t->device->context;
    set_device(d->device);
    void * p = nullptr;
    const size_t asz = size ? size : 1;
    hipError_t e = hipMalloc(&p, asz);
    if (e != hipSuccess) { (void) hipGetLastError(); MI_LOG("alloc of %.2f MiB on device %d failed: %s", size/1048576.0, d->device, hipGetErrorString(e)); return nullptr; }
    mi_buffer_ctx * c = new mi_buffer_ctx{ d->device, p, size, false };
    return make_buffer(t, g_buf_iface, c, size);
}
static size_t buft_align(ggml_backend_buffer_type_t) { return 256; }
static size_t buft_alloc_size(ggml_backend_buffer_type_t, const ggml_tensor * t) {
    // 16-byte loads of quantised rows end exactly at the row end, so no tail padding is required;
    // keep 64 spare bytes so that a vector load of the last fp32/fp16 row can never cross the allocation.
    return mi_nbytes(t) + 64;
}
static bool buft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i g_buft_iface = { buft_name, buft_alloc, buft_align, nullptr, buft_alloc_size, buft_is_host };
bool mi_buft_is_ours(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == buft_name; }

// ============================================================ pinned host buffer type
static const char * hbuft_name(ggml_backend_buffer_type_t) { return "MI355X_Host"; }
static void hbuf_memset(ggml_backend_buffer_t, ggml_tensor * t, uint8_t v, size_t off, size_t size) { memset((char *) t->data + off, v, size); }
static void hbuf_set(ggml_backend_buffer_t, ggml_tensor * t, const void * data, size_t off, size_t size) { memcpy((char *) t->data + off, data, size); }
static void hbuf_get(ggml_backend_buffer_t, const ggml_tensor * t, void * data, size_t off, size_t size) { memcpy(data, (const char *) t->data + off, size); }
static bool hbuf_cpy(ggml_backend_buffer_t, const ggml_tensor * src, ggml_tensor * dst) {
    if (src->buffer && src->buffer->buft && src->buffer->buft->iface.is_host && src->buffer->buft->iface.is_host(src->buffer->buft)) {
        memcpy(dst->data, src->data, mi_nbytes(src)); return true;
    }
    return false;
}
static void hbuf_clear(ggml_backend_buffer_t b, uint8_t v) { mi_buffer_ctx * c = (mi_buffer_ctx *) b->context; memset(c->base, v, c->size); }
static const ggml_backend_buffer_i g_hbuf_iface = { buf_free, buf_get_base, nullptr, hbuf_memset, hbuf_set, hbuf_get, hbuf_cpy, hbuf_clear, nullptr };
static ggml_backend_buffer_t hbuft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    void * p = nullptr;
    hipError_t e = hipHostMalloc(&p, size ? size : 1, hipHostMallocDefault);
    if (e != hipSuccess) { (void) hipGetLastError(); MI_LOG("pinned alloc of %.2f MiB failed: %s", size/1048576.0, hipGetErrorString(e)); return nullptr; }
    mi_buffer_ctx * c = new mi_buffer_ctx{ 0, p, size, true };
    return make_buffer(t, g_hbuf_iface, c, size);
}
static size_t hbuft_align(ggml_backend_buffer_type_t) { return 64; }
static bool hbuft_is_host(ggml_backend_buffer_type_t) { return true; }
static const ggml_backend_buffer_type_i g_hbuft_iface = { hbuft_name, hbuft_alloc, hbuft_align, nullptr, nullptr, hbuft_is_host };
bool mi_buft_is_our_host(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == hbuft_name; }

// ============================================================ backend (stream)
void * mi_scratch(mi_backend_ctx * ctx, size_t size) {
    if (size > ctx->scratch_size) {
        set_device(ctx->device);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) HIP_CHECK(hipFree(ctx->scratch));
        const size_t n = (size + (1u << 20) - 1) & ~(size_t)((1u << 20) - 1);
        HIP_CHECK(hipMalloc(&ctx->scratch, n));
        ctx->scratch_size = n;
    }
    return ctx->scratch;
}
static const char * be_name(ggml_backend_t b) { return ((mi_backend_ctx *) b->context)->name; }
static void be_free(ggml_backend_t b) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context;
    set_device(c->device);
    HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->scratch) HIP_CHECK(hipFree(c->scratch));
    if (c->act_cache) { if (c->act_cache->pool) HIP_CHECK(hipFree(c->act_cache->pool)); if (c->act_cache->big_pool) HIP_CHECK(hipFree(c->act_cache->big_pool)); if (c->act_cache->rope_tab) HIP_CHECK(hipFree(c->act_cache->rope_tab)); delete c->act_cache; }
    if (c->copy_ev) HIP_CHECK(hipEventDestroy(c->copy_ev));
    mi_split_free_events(c);
    HIP_CHECK(hipStreamDestroy(c->stream));
    delete c; delete b;
}
static void be_set_async(ggml_backend_t b, ggml_tensor * t, const void * data, size_t off, size_t size) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    before_write(t, off, size);
    HIP_CHECK(hipMemcpyAsync((char *) t->data + off, data, size, hipMemcpyHostToDevice, c->stream));
}
static void be_get_async(ggml_backend_t b, const ggml_tensor * t, void * data, size_t off, size_t size) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    if (mi_is_tiled(tile_root(t))) { tiled_read(t, data, off, size); return; }
    HIP_CHECK(hipMemcpyAsync(data, (const char *) t->data + off, size, hipMemcpyDeviceToHost, c->stream));
}
static bool be_is_ours(ggml_backend_t b);
static bool be_cpy_async(ggml_backend_t bs, ggml_backend_t bd, const ggml_tensor * src, ggml_tensor * dst) {
    if (!be_is_ours(bs) || !be_is_ours(bd)) return false;
    if (!mi_buffer_is_ours(src->buffer) || !mi_buffer_is_ours(dst->buffer)) return false;
    if (mi_is_tiled(tile_root(src))) return false;
    before_write(dst, 0, mi_nbytes(dst));
    mi_backend_ctx * cs = (mi_backend_ctx *) bs->context, * cd = (mi_backend_ctx *) bd->context;
    const size_t n = mi_nbytes(dst);
    if (bs == bd) {
        set_device(cd->device);
        HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, cd->stream));
        return true;
    }
    // copy on the source stream, then make the destination stream wait for it (same protocol as the reference :2350-2403)
    set_device(cs->device);
    if (cs->device == cd->device) HIP_CHECK(hipMemcpyAsync(dst->data, src->data, n, hipMemcpyDeviceToDevice, cs->stream));
    else                          HIP_CHECK(hipMemcpyPeerAsync(dst->data, cd->device, src->data, cs->device, n, cs->stream));
    if (!cs->copy_ev) HIP_CHECK(hipEventCreateWithFlags(&cs->copy_ev, hipEventDisableTiming));      // one event per source backend, re-recorded per copy
    HIP_CHECK(hipEventRecord(cs->copy_ev, cs->stream));
    set_device(cd->device);
    HIP_CHECK(hipStreamWaitEvent(cd->stream, cs->copy_ev, 0));
    return true;
}
void mi_allow_big_lds(const void * fn) {
    static std::mutex mu; static std::vector<std::pair<const void *, int>> done;
    int dev = 0; HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    for (const auto & d : done) if (d.first == fn && d.second == dev) return;
    HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
    done.emplace_back(fn, dev);
}
static void be_sync(ggml_backend_t b) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    // poll the stream for up to GGML_MI355X_SYNC_SPIN us (default 4000) before falling back to the blocking wait: the wake-up of a
    // blocked host thread is part of every decode's latency (the reference's CUDA backend spins inside cudaStreamSynchronize by default)
    static const long spin_us = [] { const char * e = getenv("GGML_MI355X_SYNC_SPIN"); return e ? atol(e) : 4000L; }();
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const hipError_t q = hipStreamQuery(c->stream);
            if (q == hipSuccess) return;
            if (q != hipErrorNotReady) { HIP_CHECK(q); return; }
            if (std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count() > spin_us) break;
        }
    }
    HIP_CHECK(hipStreamSynchronize(c->stream));
}
static enum ggml_status be_graph_compute(ggml_backend_t b, ggml_cgraph * g) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    return mi_graph_compute(c, g);
}
static void be_event_record(ggml_backend_t b, ggml_backend_event_t e) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    HIP_CHECK(hipEventRecord((hipEvent_t) e->context, c->stream));
}
static void be_event_wait(ggml_backend_t b, ggml_backend_event_t e) {
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    HIP_CHECK(hipStreamWaitEvent(c->stream, (hipEvent_t) e->context, 0));
}
static const ggml_backend_i g_be_iface = {
    be_name, be_free, be_set_async, be_get_async, be_cpy_async, be_sync,
    nullptr, nullptr, nullptr, nullptr, be_graph_compute, be_event_record, be_event_wait
};
static bool be_is_ours(ggml_backend_t b) { return b && b->iface.get_name == be_name; }

// ============================================================ device
static const char * dev_name(ggml_backend_dev_t d) { return ((mi_device_ctx *) d->context)->name; }
static const char * dev_desc(ggml_backend_dev_t d) { return ((mi_device_ctx *) d->context)->desc; }
static void dev_memory(ggml_backend_dev_t d, size_t * fr, size_t * tot) {
    set_device(((mi_device_ctx *) d->context)->device);
    HIP_CHECK(hipMemGetInfo(fr, tot));
}
static enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
static void dev_props(ggml_backend_dev_t d, ggml_backend_dev_props * p) {
    p->name = dev_name(d); p->description = dev_desc(d); p->type = GGML_BACKEND_DEVICE_TYPE_GPU;
    dev_memory(d, &p->memory_free, &p->memory_total);
    p->caps.async = true; p->caps.host_buffer = true; p->caps.buffer_from_host_ptr = false; p->caps.events = true;
}
static ggml_backend_t dev_init_backend(ggml_backend_dev_t d, const char *) {
    mi_device_ctx * dc = (mi_device_ctx *) d->context;
    set_device(dc->device);
    mi_backend_ctx * c = new mi_backend_ctx{};
    c->device = dc->device;
    HIP_CHECK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    snprintf(c->name, sizeof(c->name), "%s", dc->name);
    c->act_cache = new mi_act_cache;
    c->act_cache->slot_bytes = 512*1024;                                   // T = 8, k = 28672 (70B ffn_down) needs 262 KB
    HIP_CHECK(hipMalloc((void **) &c->act_cache->pool, c->act_cache->slot_bytes * MI_ACT_SLOTS));
    ggml_backend * b = new ggml_backend;
    b->guid = &g_guid; b->iface = g_be_iface; b->device = d; b->context = c;
    return b;
}
static ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t d) { return &g_bufts[d - g_devs]; }
static ggml_backend_buffer_type_t dev_host_buft(ggml_backend_dev_t) { return &g_host_buft; }
static bool dev_supports_op(ggml_backend_dev_t d, const ggml_tensor * op) { return mi_supports_op(((mi_device_ctx *) d->context)->device, op); }
static bool dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t t) {
    // buffers of this device, usable from any backend instance (target and draft contexts share weights)
    return (mi_buft_is_ours(t) || mi_buft_is_split(t)) && t->device == d;
}
static bool dev_offload_op(ggml_backend_dev_t, const ggml_tensor * op) {
    const int min_batch = 32;                                          // same threshold as the reference :3250-3256
    return (op->ne[1] >= min_batch && op->op != GGML_OP_GET_ROWS) || (op->ne[2] >= min_batch && op->op == GGML_OP_MUL_MAT_ID);
}
static ggml_backend_event_t dev_event_new(ggml_backend_dev_t d) {
    set_device(((mi_device_ctx *) d->context)->device);
    hipEvent_t ev; HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    return new ggml_backend_event{ d, ev };
}
static void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t e) { HIP_CHECK(hipEventDestroy((hipEvent_t) e->context)); delete e; }
static void dev_event_sync(ggml_backend_dev_t, ggml_backend_event_t e) { HIP_CHECK(hipEventSynchronize((hipEvent_t) e->context)); }
static const ggml_backend_device_i g_dev_iface = {
    dev_name, dev_desc, dev_memory, dev_type, dev_props, dev_init_backend, dev_buft, dev_host_buft, nullptr,
    dev_supports_op, dev_supports_buft, dev_offload_op, dev_event_new, dev_event_free, dev_event_sync
};

// ============================================================ registry
static const char * reg_name(ggml_backend_reg_t) { return "MI355X"; }
static size_t reg_dev_count(ggml_backend_reg_t) { return (size_t) g_ndev; }
static ggml_backend_dev_t reg_dev_get(ggml_backend_reg_t, size_t i) { MI_ASSERT((int) i < g_ndev); return &g_devs[i]; }

static ggml_backend_feature g_features[] = {
    { "ARCH", "gfx950" }, { "WAVE", "64" }, { "ACT_QUANT", "cpu-parity(q8_K/q8_0)" }, { "GRAPH_FUSION", "1" }, { nullptr, nullptr },
};
static ggml_backend_feature * reg_get_features(ggml_backend_reg_t) { return g_features; }

// row-split weights: implemented in split.cpp (tensor-parallel over the visible devices)
extern "C" ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split);

// extension for hosts that interleave their own stream work (RCCL collectives between graph segments)
static void * backend_stream(ggml_backend_t b) { return be_is_ours(b) ? (void *) ((mi_backend_ctx *) b->context)->stream : nullptr; }
// extension for hosts that draft trees: the k best logits of a few rows without bringing the rows to the host (include/ggml_mi355x.h)
extern "C" GGML_MI355X_API int ggml_backend_mi355x_top_k(ggml_backend_t b, const ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals) {
    if (!be_is_ours(b) || !logits || !logits->buffer || !mi_buffer_is_ours(logits->buffer) || !logits->data || n_rows < 0 || !mi_top_k_supported(logits, k)) return -1;
    if (n_rows == 0) return 0;
    mi_backend_ctx * c = (mi_backend_ctx *) b->context; set_device(c->device);
    for (int i = 0; i < n_rows; ++i) if ((rows ? rows[i] : i) < 0 || (rows ? rows[i] : i) >= logits->ne[1]) return -1;
    // results: [n_rows][k] ids | [n_rows][k] values in a small device buffer and its page-locked twin, kept with the backend instance
    static std::mutex mu; struct stage { mi_backend_ctx * ctx; char * dev; char * host; size_t cap; }; static std::vector<stage> stages;
    const size_t need = (size_t) n_rows * k * 8;
    stage * s = nullptr;
    { std::lock_guard<std::mutex> lk(mu); for (auto & e : stages) if (e.ctx == c) s = &e; if (!s) { stages.push_back({ c, nullptr, nullptr, 0 }); s = &stages.back(); }
      if (s->cap < need) {
          HIP_CHECK(hipStreamSynchronize(c->stream));
          if (s->dev) HIP_CHECK(hipFree(s->dev)); if (s->host) HIP_CHECK(hipHostFree(s->host));
          s->cap = need < 16384 ? 16384 : need * 2;
          HIP_CHECK(hipMalloc((void **) &s->dev, s->cap)); HIP_CHECK(hipHostMalloc((void **) &s->host, s->cap, hipHostMallocDefault));
      } }
    int32_t * d_ids = (int32_t *) s->dev; float * d_vals = (float *)(s->dev + (size_t) n_rows * k * 4);
    for (int r0 = 0; r0 < n_rows; r0 += 16) {
        const int n = n_rows - r0 < 16 ? n_rows - r0 : 16;
        int32_t rr[16]; for (int i = 0; i < n; ++i) rr[i] = rows ? rows[r0 + i] : r0 + i;
        mi_top_k(c->stream, logits, rr, n, k, d_ids + (size_t) r0 * k, d_vals + (size_t) r0 * k);
    }
    HIP_CHECK(hipMemcpyAsync(s->host, s->dev, need, hipMemcpyDeviceToHost, c->stream));
    HIP_CHECK(hipStreamSynchronize(c->stream));
    memcpy(ids, s->host, (size_t) n_rows * k * 4); memcpy(vals, s->host + (size_t) n_rows * k * 4, (size_t) n_rows * k * 4);
    return 0;
}
// hosts that need a collective between two nodes of a graph (tensor parallel: all-reduce of the partial sums after wo / ffn_down):
// `fn(user, t, stream)` is called from inside graph_compute when the node that produces nodes[i] has been queued on `stream`; what fn
// enqueues there (ncclAllReduce in place on t->data) runs before every later node.  The tensors should carry GGML_TENSOR_FLAG_OUTPUT
// so that no fusion swallows them.  nodes == NULL / n == 0 removes the hooks.  The array must stay valid while they are set.
extern "C" GGML_MI355X_API int ggml_backend_mi355x_set_node_hooks(ggml_backend_t b, const ggml_tensor * const * nodes, int n,
                                                                  void (*fn)(void * user, const ggml_tensor * t, void * stream), void * user) {
    if (!be_is_ours(b)) return -1;
    mi_backend_ctx * c = (mi_backend_ctx *) b->context;
    if (!nodes || n <= 0 || !fn) { c->node_hook = nullptr; c->node_hook_user = nullptr; c->hook_nodes = nullptr; c->n_hook_nodes = 0; return 0; }
    c->node_hook = fn; c->node_hook_user = user; c->hook_nodes = nodes; c->n_hook_nodes = n;
    return 0;
}
static void * reg_proc(ggml_backend_reg_t, const char * name) {
    if (!strcmp(name, "ggml_backend_mi355x_set_node_hooks")) return (void *) ggml_backend_mi355x_set_node_hooks;
    if (!strcmp(name, "ggml_backend_mi355x_stream"))     return (void *) backend_stream;
    if (!strcmp(name, "ggml_backend_mi355x_top_k"))      return (void *) ggml_backend_mi355x_top_k;
    if (!strcmp(name, "ggml_backend_get_features"))      return (void *) reg_get_features;
    if (!strcmp(name, "ggml_backend_split_buffer_type")) return (void *) ggml_backend_mi355x_split_buffer_type;
    return nullptr;
}
static const ggml_backend_reg_i g_reg_iface = { reg_name, reg_dev_count, reg_dev_get, reg_proc };

static void init_once() {
    static std::once_flag once;
    std::call_once(once, [] {
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess) { (void) hipGetLastError(); n = 0; }
        if (n > MI_MAX_DEVICES) n = MI_MAX_DEVICES;
        g_reg.api_version = GGML_BACKEND_API_VERSION; g_reg.iface = g_reg_iface; g_reg.context = nullptr;
        int kept = 0;
        for (int i = 0; i < n; ++i) {
            hipDeviceProp_t p;
            if (hipGetDeviceProperties(&p, i) != hipSuccess) { (void) hipGetLastError(); continue; }
            // code objects are gfx950 only: refuse anything else instead of failing at first launch
            if (strncmp(p.gcnArchName, "gfx950", 6) != 0) { MI_LOG("device %d is %s, not gfx950 -- skipped", i, p.gcnArchName); continue; }
            mi_device_ctx & d = g_devctx[kept];              // all three tables by the LOGICAL index (kept devices); d.device = the HIP ordinal
            d.device = i;
            snprintf(d.name, sizeof(d.name), "MI355X%d", i);
            snprintf(d.desc, sizeof(d.desc), "%s (%s, %d CUs, %.0f GiB)", p.name, p.gcnArchName, p.multiProcessorCount, p.totalGlobalMem / 1073741824.0);
            g_devs[kept].iface = g_dev_iface; g_devs[kept].reg = &g_reg; g_devs[kept].context = &d;
            g_bufts[kept].iface = g_buft_iface; g_bufts[kept].device = &g_devs[kept]; g_bufts[kept].context = nullptr;
            ++kept;
        }
        g_ndev = kept;
        g_host_buft.iface = g_hbuft_iface; g_host_buft.device = kept ? &g_devs[0] : nullptr; g_host_buft.context = nullptr;
    });
}

extern "C" {
GGML_MI355X_API ggml_backend_reg_t ggml_backend_init(void) { init_once(); return &g_reg; }
GGML_MI355X_API int ggml_backend_score(void) { init_once(); return g_ndev > 0 ? 100 : 0; }
GGML_MI355X_API ggml_backend_reg_t ggml_backend_mi355x_reg(void) { init_once(); return &g_reg; }
GGML_MI355X_API int ggml_backend_mi355x_device_count(void) { init_once(); return g_ndev; }
}
