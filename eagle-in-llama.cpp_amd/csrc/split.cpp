// split.cpp -- row-split (tensor-parallel, `-sm row`) weight buffer type and the MUL_MAT that reads it.
//
// Reference: R/ggml/src/ggml-cuda/ggml-cuda.cu
//   :720-748   get_row_rounding / get_row_split: device id owns rows [nrows*split[id], nrows*split[id+1]) rounded down, last device to nrows
//   :750-1046  ggml_backend_cuda_split_buffer_type / buffer: per-device allocations hung off tensor->extra, set_tensor slices rows,
//              get_tensor gathers, no base pointer (a fake one), supports only 2-D weights
//   :1359-1667 ggml_cuda_op_mul_mat, split branch: activations quantised on the main device, peer-copied to every device, each runs its
//              row slice, dst slices come back with a 2-D peer copy; everything joined with events on the main stream
//   caller     R/src/llama-model.cpp:310-322 (get_proc_address("ggml_backend_split_buffer_type")), weight placement through supports_op
//
// Here (MI355X-first): the fp32 activations (T x k x 4 bytes -- 96 KB for a 6-token verification) go to every device over xGMI and each
// device quantises them in the prologue of its own mat-vec launch (kernels_mmt.hip), so no quantise-then-broadcast dependency exists.
// With peer access (enabled once per device pair) the epilogue of a remote slice stores its dst rows straight into the main device's
// dst over xGMI: no gather copy at all; without it the slice crosses as one contiguous peer copy and the main stream scatters it with
// one 2-D copy.  Slices are re-laid out into 16-row tiles (tile_layout.h) as they are uploaded.  One HIP stream per (logical) device
// with its own lock; the events that join them belong to the calling backend instance and live as long as it does.
// GGML_MI355X_SPLIT_FAKE_DEVICES=N makes the type treat ONE physical GPU as N logical devices (separate slices, streams, staging):
// the whole path -- slicing, broadcast, gather, joins -- is then exercised on a single-GPU box (tests/test_split_gpu.py).
#include "mi355x_common.h"
#include "ggml_mi355x.h"
#include "kernels.h"
#include "tile_layout.h"
#include <mutex>
#include <vector>
#include <string>
#include <cmath>

#define SPLIT_MAX 16
#define SPLIT_ROUNDING 128           // row boundaries: multiples of 128 (the reference rounds to its mmq tile height; ours must keep 16-row tiles whole)

static int fake_devices() { static const int n = [] { const char * e = getenv("GGML_MI355X_SPLIT_FAKE_DEVICES"); const int v = e ? atoi(e) : 0; return v > SPLIT_MAX ? SPLIT_MAX : v; }(); return n; }
static int n_logical() { const int f = fake_devices(); return f > 0 ? f : mi_device_count(); }
static int logical_dev(int d) { return fake_devices() > 0 ? 0 : d; }          // index into the plugin's device table
static int phys(int d) { return mi_device_ordinal(logical_dev(d)); }            // HIP ordinal (a skipped non-gfx950 device may precede a kept one)

// cumulative fractions -> row range of device id (get_row_split, ggml-cuda.cu:735-748)
static void row_split(int64_t nrows, const float * cum, int n_dev, int id, int64_t * lo, int64_t * hi) {
    int64_t l = id == 0 ? 0 : (int64_t)(nrows * cum[id]);
    l -= l % SPLIT_ROUNDING;
    int64_t h;
    if (id == n_dev - 1) h = nrows; else { h = (int64_t)(nrows * cum[id + 1]); h -= h % SPLIT_ROUNDING; }
    *lo = l; *hi = h < l ? l : h;
}
// tensor_split (per-device proportions; NULL or all zero = even) -> cumulative start fractions (ggml_backend_cuda_split_buffer_type :1003-1023)
static void cumulative(const float * tensor_split, int n_dev, float * cum) {
    float sum = 0.f;
    bool all_zero = tensor_split == nullptr;
    if (!all_zero) { all_zero = true; for (int i = 0; i < n_dev; ++i) if (tensor_split[i] != 0.0f) all_zero = false; }
    float acc = 0.f;
    for (int i = 0; i < n_dev; ++i) sum += all_zero ? 1.0f : tensor_split[i];
    for (int i = 0; i < n_dev; ++i) { cum[i] = acc / sum; acc += all_zero ? 1.0f : tensor_split[i]; }
}
// exported for the CPU-side unit test of the slicing arithmetic (no GPU needed)
extern "C" GGML_MI355X_API void ggml_backend_mi355x_row_split(int64_t nrows, const float * tensor_split, int n_dev, int id, int64_t * lo, int64_t * hi) {
    float cum[SPLIT_MAX]; if (n_dev > SPLIT_MAX) n_dev = SPLIT_MAX;
    cumulative(tensor_split, n_dev, cum);
    row_split(nrows, cum, n_dev, id, lo, hi);
}

struct split_buft_ctx { int main_device; int n_dev; float cum[SPLIT_MAX]; std::string name; };
struct split_extra { void * data[SPLIT_MAX] = {}; int64_t lo[SPLIT_MAX] = {}, hi[SPLIT_MAX] = {}; bool tiled[SPLIT_MAX] = {}; };
struct split_buf_ctx { split_buft_ctx * bt; std::vector<split_extra *> extras; };

// ---- per logical device: stream, staging, activation-image cache.  Launches on one device are serialised by ITS mutex (held while the
// work of one MUL_MAT is queued on that device's stream, not across devices); events belong to the calling backend instance (below).
struct split_dev {
    std::mutex mu;
    hipStream_t stream = nullptr;
    char * xs = nullptr; size_t xs_cap = 0;           // activations of the current MUL_MAT (on this device)
    char * ys = nullptr; size_t ys_cap = 0;           // staged path only: the slice's output [T][n] (on this device)
    char * hs = nullptr; size_t hs_cap = 0; int hs_dev = -1;      // staged path only: its twin on the main device
    mi_act_cache cache; void * tmp = nullptr; size_t tmp_cap = 0;
};
static split_dev g_sd[SPLIT_MAX];
static std::mutex g_split_init_mu;                    // creation of the per-device objects and the peer table only
static split_dev & sdev(int d) {
    split_dev & s = g_sd[d];
    std::lock_guard<std::mutex> lk(g_split_init_mu);
    if (!s.stream) {
        HIP_CHECK(hipSetDevice(phys(d)));
        HIP_CHECK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        s.cache.slot_bytes = 512*1024;
        HIP_CHECK(hipMalloc((void **) &s.cache.pool, s.cache.slot_bytes * MI_ACT_SLOTS));
    }
    return s;
}
// staging grows rarely (first cap 4 MB = a 128-token batch of 8192 floats): only the stream that uses the buffer is drained first
static void grow(char *& p, size_t & cap, size_t n, int hip_dev, hipStream_t user) {
    if (n <= cap) return;
    HIP_CHECK(hipSetDevice(hip_dev));
    if (p) { HIP_CHECK(hipStreamSynchronize(user)); HIP_CHECK(hipFree(p)); }
    cap = n < (4u << 20) ? (4u << 20) : ((n + (n >> 1) + (1u << 20)) & ~(size_t)((1u << 20) - 1));
    HIP_CHECK(hipMalloc((void **) &p, cap));
}
// peer access, enabled once per ordered device pair (reference: ggml_cuda_set_peer_access, ggml-cuda.cu:1281-1337): `from` may then
// dereference `to`'s memory -- the mat-vec epilogue of a remote slice stores its dst rows straight into the main device's dst over xGMI
static bool peer_ok(int from, int to) {
    static int8_t state[MI_MAX_DEVICES][MI_MAX_DEVICES] = {};          // 0 unknown, 1 enabled, -1 not possible
    if (from == to) return true;
    if (from < 0 || to < 0 || from >= MI_MAX_DEVICES || to >= MI_MAX_DEVICES) return false;
    std::lock_guard<std::mutex> lk(g_split_init_mu);
    if (state[from][to] == 0) {
        int can = 0;
        HIP_CHECK(hipSetDevice(from));
        if (hipDeviceCanAccessPeer(&can, from, to) != hipSuccess) { (void) hipGetLastError(); can = 0; }
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(to, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) can = 0;
            (void) hipGetLastError();
        }
        state[from][to] = can ? 1 : -1;
        if (!can) MI_LOG("split: device %d cannot map device %d's memory -- slices on it return through a staged copy", from, to);
    }
    return state[from][to] > 0;
}
// events of one backend instance (a backend is used by one thread at a time): `ready` on its own device, done[d] on logical device d
struct mi_split_events { hipEvent_t ready = nullptr, scattered = nullptr; hipEvent_t done[SPLIT_MAX] = {}; };
static mi_split_events * events_of(mi_backend_ctx * ctx) {
    if (!ctx->split_ev) {
        mi_split_events * ev = new mi_split_events;
        HIP_CHECK(hipSetDevice(ctx->device));
        HIP_CHECK(hipEventCreateWithFlags(&ev->ready, hipEventDisableTiming));
        HIP_CHECK(hipEventCreateWithFlags(&ev->scattered, hipEventDisableTiming));
        ctx->split_ev = ev;
    }
    return ctx->split_ev;
}
void mi_split_free_events(mi_backend_ctx * ctx) {
    mi_split_events * ev = ctx->split_ev;
    if (!ev) return;
    if (ev->ready) HIP_CHECK(hipEventDestroy(ev->ready));
    if (ev->scattered) HIP_CHECK(hipEventDestroy(ev->scattered));
    for (int d = 0; d < SPLIT_MAX; ++d) if (ev->done[d]) HIP_CHECK(hipEventDestroy(ev->done[d]));
    delete ev; ctx->split_ev = nullptr;
}

// ============================================================ buffer
static void sbuf_free(ggml_backend_buffer_t b) {
    split_buf_ctx * c = (split_buf_ctx *) b->context;
    for (split_extra * e : c->extras) { for (int d = 0; d < c->bt->n_dev; ++d) if (e->data[d]) { HIP_CHECK(hipSetDevice(phys(d))); HIP_CHECK(hipFree(e->data[d])); } delete e; }
    delete c;
}
static void * sbuf_base(ggml_backend_buffer_t) { return (void *) 0x1000; }      // never dereferenced: the data lives in the per-device slices (reference :826-830)
static bool split_ok_tensor(const ggml_tensor * t) { return t->ne[2] == 1 && t->ne[3] == 1 && !t->view_src && mi_traits(t->type).blck > 0 && t->nb[1] == mi_row_size(t->type, t->ne[0]); }
static void sbuf_init_tensor(ggml_backend_buffer_t b, ggml_tensor * t) {
    split_buf_ctx * c = (split_buf_ctx *) b->context;
    MI_ASSERT(split_ok_tensor(t));                                                // 2-D, whole rows (reference :834)
    split_extra * e = new split_extra; c->extras.push_back(e);
    const size_t rb = mi_row_size(t->type, t->ne[0]);
    for (int d = 0; d < c->bt->n_dev; ++d) {
        row_split(t->ne[1], c->bt->cum, c->bt->n_dev, d, &e->lo[d], &e->hi[d]);
        const int64_t n = e->hi[d] - e->lo[d];
        if (n <= 0) continue;
        HIP_CHECK(hipSetDevice(phys(d)));
        HIP_CHECK(hipMalloc(&e->data[d], (size_t) n * rb + 256));
    }
    t->extra = e;
}
static void sbuf_set_tensor(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t off, size_t size) {
    split_buf_ctx * c = (split_buf_ctx *) b->context;
    MI_ASSERT(off == 0 && size == mi_nbytes(t));                                  // split tensors are set whole (reference :846-848)
    split_extra * e = (split_extra *) t->extra;
    const size_t rb = mi_row_size(t->type, t->ne[0]);
    for (int d = 0; d < c->bt->n_dev; ++d) {
        const int64_t n = e->hi[d] - e->lo[d];
        if (n <= 0) continue;
        split_dev & s = sdev(d);
        std::lock_guard<std::mutex> lk(s.mu);
        HIP_CHECK(hipSetDevice(phys(d)));
        const char * src = (const char *) data + (size_t) e->lo[d] * rb;
        const bool tile = mi_unit_bytes(t->type) > 0 && n % 16 == 0 && t->ne[0] % 256 == 0 && getenv("GGML_MI355X_NO_TILE") == nullptr;
        if (tile) {     // upload next to the slice, permute into it (tile_layout.h)
            grow((char *&) s.tmp, s.tmp_cap, (size_t) n * rb, phys(d), s.stream);
            HIP_CHECK(hipMemcpyAsync(s.tmp, src, (size_t) n * rb, hipMemcpyHostToDevice, s.stream));
            mi_tile_permute(s.stream, s.tmp, e->data[d], t->type, n, t->ne[0], true);
        } else HIP_CHECK(hipMemcpyAsync(e->data[d], src, (size_t) n * rb, hipMemcpyHostToDevice, s.stream));
        e->tiled[d] = tile;
        HIP_CHECK(hipStreamSynchronize(s.stream));
    }
}
static void sbuf_get_tensor(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t off, size_t size) {
    split_buf_ctx * c = (split_buf_ctx *) b->context;
    MI_ASSERT(off == 0 && size == mi_nbytes(t));
    split_extra * e = (split_extra *) t->extra;
    const size_t rb = mi_row_size(t->type, t->ne[0]);
    for (int d = 0; d < c->bt->n_dev; ++d) {
        const int64_t n = e->hi[d] - e->lo[d];
        if (n <= 0) continue;
        split_dev & s = sdev(d);
        std::lock_guard<std::mutex> lk(s.mu);
        HIP_CHECK(hipSetDevice(phys(d)));
        const void * src = e->data[d];
        if (e->tiled[d]) { grow((char *&) s.tmp, s.tmp_cap, (size_t) n * rb, phys(d), s.stream); mi_tile_permute(s.stream, e->data[d], s.tmp, t->type, n, t->ne[0], false); src = s.tmp; }
        HIP_CHECK(hipMemcpyAsync((char *) data + (size_t) e->lo[d] * rb, src, (size_t) n * rb, hipMemcpyDeviceToHost, s.stream));
        HIP_CHECK(hipStreamSynchronize(s.stream));
    }
}
static void sbuf_clear(ggml_backend_buffer_t, uint8_t) {}
static const ggml_backend_buffer_i g_sbuf_iface = { sbuf_free, sbuf_base, sbuf_init_tensor, nullptr, sbuf_set_tensor, sbuf_get_tensor, nullptr, sbuf_clear, nullptr };

// ============================================================ buffer type
static const char * sbuft_name(ggml_backend_buffer_type_t t) { return ((split_buft_ctx *) t->context)->name.c_str(); }
static ggml_backend_buffer_t sbuft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    // the sizes the allocator adds up (get_alloc_size) are only bookkeeping: slices are allocated per tensor in init_tensor (reference :938-947)
    split_buf_ctx * c = new split_buf_ctx{ (split_buft_ctx *) t->context, {} };
    return mi_make_buffer(t, g_sbuf_iface, c, size);
}
static size_t sbuft_align(ggml_backend_buffer_type_t) { return 128; }
static size_t sbuft_alloc_size(ggml_backend_buffer_type_t t, const ggml_tensor * x) {
    split_buft_ctx * c = (split_buft_ctx *) t->context;
    size_t total = 0; const size_t rb = mi_row_size(x->type, x->ne[0]);
    for (int d = 0; d < c->n_dev; ++d) { int64_t lo, hi; row_split(x->ne[1], c->cum, c->n_dev, d, &lo, &hi); if (hi > lo) total += (size_t)(hi - lo) * rb + 256; }
    return total;
}
static bool sbuft_is_host(ggml_backend_buffer_type_t) { return false; }
static const ggml_backend_buffer_type_i g_sbuft_iface = { sbuft_name, sbuft_alloc, sbuft_align, nullptr, sbuft_alloc_size, sbuft_is_host };
bool mi_buft_is_split(ggml_backend_buffer_type_t buft) { return buft && buft->iface.get_name == sbuft_name; }
bool mi_tensor_is_split(const ggml_tensor * t) { return t && t->buffer && mi_buft_is_split(t->buffer->buft); }

extern "C" GGML_MI355X_API ggml_backend_buffer_type_t ggml_backend_mi355x_split_buffer_type(int main_device, const float * tensor_split) {
    static std::mutex mu;
    static std::vector<ggml_backend_buffer_type *> made;                            // one object per distinct (main device, split), process lifetime (reference :1025-1040)
    std::lock_guard<std::mutex> lk(mu);
    const int n = n_logical();
    if (n < 1 || main_device < 0 || main_device >= n || !mi_device(logical_dev(main_device))) return nullptr;
    split_buft_ctx want; want.main_device = main_device; want.n_dev = n; cumulative(tensor_split, n, want.cum);
    for (auto * bt : made) { split_buft_ctx * c = (split_buft_ctx *) bt->context; if (c->main_device == main_device && c->n_dev == n && !memcmp(c->cum, want.cum, sizeof(float) * n)) return bt; }
    split_buft_ctx * c = new split_buft_ctx(want);
    c->name = "MI355X_Split";
    ggml_backend_buffer_type * bt = new ggml_backend_buffer_type{ g_sbuft_iface, mi_device(logical_dev(main_device)), c };
    made.push_back(bt);
    return bt;
}

// ============================================================ MUL_MAT over a split weight
bool mi_split_supports_mul_mat(const ggml_tensor * op) {
    const ggml_tensor * w = op->src[0], * x = op->src[1];
    if (!w || !x || !mi_tensor_is_split(w) || !mi_mul_mat_q_supported_type(w->type)) return false;
    if (x->type != GGML_TYPE_F32 || op->type != GGML_TYPE_F32 || x->ne[2] != 1 || x->ne[3] != 1 || x->nb[0] != 4 || op->nb[0] != 4 || (x->nb[1] % 16)) return false;
    const auto tr = mi_traits(w->type);
    return w->ne[0] % tr.blck == 0 && w->ne[0] <= 128*1024 && !mi_tensor_is_split(x);
}
void mi_split_mul_mat(mi_backend_ctx * ctx, const ggml_tensor * dst) {
    const ggml_tensor * w = dst->src[0], * x = dst->src[1];
    split_buft_ctx * bt = (split_buft_ctx *) w->buffer->buft->context;
    split_extra * e = (split_extra *) w->extra;
    const int k = (int) w->ne[0], T = (int) x->ne[1];
    if (T == 0 || k == 0 || w->ne[1] == 0) return;
    static const bool force_stage = getenv("GGML_MI355X_SPLIT_STAGE") != nullptr;    // tests: take the no-peer-access path on any box
    const int home = ctx->device;                                                  // HIP ordinal of the calling backend: dst and x live there
    mi_split_events * ev = events_of(ctx);
    // everything queued on the caller's stream so far (x among it) must be visible to the other devices' streams
    HIP_CHECK(hipSetDevice(home)); HIP_CHECK(hipEventRecord(ev->ready, ctx->stream));
    const size_t xbytes = (size_t) T * x->nb[1];
    struct scatter { int d; int64_t lo, n; } late[SPLIT_MAX]; int n_late = 0;
    for (int d = 0; d < bt->n_dev; ++d) {
        const int64_t lo = e->lo[d], n = e->hi[d] - e->lo[d];
        if (n <= 0) continue;
        split_dev & s = sdev(d);
        const int pd = phys(d);
        std::lock_guard<std::mutex> lk(s.mu);
        HIP_CHECK(hipSetDevice(pd));
        if (!ev->done[d]) HIP_CHECK(hipEventCreateWithFlags(&ev->done[d], hipEventDisableTiming));
        HIP_CHECK(hipStreamWaitEvent(s.stream, ev->ready, 0));
        const bool local = pd == home && fake_devices() == 0;                      // the slice on the caller's own device reads x in place
        const bool direct = local || (!force_stage && peer_ok(pd, home));          // dst rows stored by the epilogue itself
        const float * xd = (const float *) x->data;
        if (!local) {    // one copy of the activations per device (T x k floats; the reference sends its q8_1 image, :1590-1592)
            grow(s.xs, s.xs_cap, xbytes + 64, pd, s.stream);
            if (pd == home) HIP_CHECK(hipMemcpyAsync(s.xs, x->data, xbytes, hipMemcpyDeviceToDevice, s.stream));
            else            HIP_CHECK(hipMemcpyPeerAsync(s.xs, pd, x->data, home, xbytes, s.stream));
            xd = (const float *) s.xs;
        }
        char * yd = (char *) dst->data + (size_t) lo * 4; int64_t y_tok = dst->nb[1];
        if (!direct) { grow(s.ys, s.ys_cap, (size_t) T * n * 4 + 64, pd, s.stream); yd = s.ys; y_tok = n * 4; }
        s.cache.epoch++;
        mmvq_launch L{};
        L.act.X = xd; L.act.xs = x->nb[1] / 4; L.k = k; L.n_mat = 1; L.tiled = e->tiled[d] ? 1 : 0;
        L.m[0].W = (const char *) e->data[d]; L.m[0].row_bytes = (int64_t) mi_row_size(w->type, k); L.m[0].rows = (int) n;
        L.m[0].epi = EPI_F32; L.m[0].out = yd; L.m[0].o_row = 4; L.m[0].o_tok = y_tok;
        mi_mmvq_run(s.stream, w->type, T, L, &s.cache, nullptr);
        if (!direct) {   // no mapping of the main device's memory: the slice's [T][n] block crosses as ONE contiguous copy, the main stream scatters it
            if (s.hs_dev != home) { if (s.hs) { HIP_CHECK(hipSetDevice(s.hs_dev)); HIP_CHECK(hipStreamSynchronize(s.stream)); HIP_CHECK(hipFree(s.hs)); HIP_CHECK(hipSetDevice(pd)); } s.hs = nullptr; s.hs_cap = 0; s.hs_dev = home; }
            grow(s.hs, s.hs_cap, (size_t) T * n * 4 + 64, home, s.stream);
            HIP_CHECK(hipSetDevice(pd));
            if (pd == home) HIP_CHECK(hipMemcpyAsync(s.hs, s.ys, (size_t) T * n * 4, hipMemcpyDeviceToDevice, s.stream));
            else            HIP_CHECK(hipMemcpyPeerAsync(s.hs, home, s.ys, pd, (size_t) T * n * 4, s.stream));
            late[n_late++] = { d, lo, n };
        }
        HIP_CHECK(hipEventRecord(ev->done[d], s.stream));
    }
    HIP_CHECK(hipSetDevice(home));
    for (int d = 0; d < bt->n_dev; ++d) if (e->hi[d] > e->lo[d]) HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev->done[d], 0));       // join (reference :1653-1666)
    // dst[t][lo .. lo+n) <- hs[t][0 .. n): one 2-D copy per staged slice (reference :1631-1635), on the main device's own stream
    for (int i = 0; i < n_late; ++i)
        HIP_CHECK(hipMemcpy2DAsync((char *) dst->data + (size_t) late[i].lo * 4, dst->nb[1], g_sd[late[i].d].hs, (size_t) late[i].n * 4, (size_t) late[i].n * 4, T, hipMemcpyDeviceToDevice, ctx->stream));
    if (n_late) {
        // the staging twins are re-used by the next MUL_MAT on these devices -- possibly issued through ANOTHER backend instance, whose stream is
        // not ordered with this one: the devices' streams wait until the scatters above have read them
        HIP_CHECK(hipEventRecord(ev->scattered, ctx->stream));
        for (int i = 0; i < n_late; ++i) { split_dev & s = g_sd[late[i].d]; std::lock_guard<std::mutex> lk(s.mu); HIP_CHECK(hipSetDevice(phys(late[i].d))); HIP_CHECK(hipStreamWaitEvent(s.stream, ev->scattered, 0)); }
        HIP_CHECK(hipSetDevice(home));
    }
}
