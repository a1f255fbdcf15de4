// lab_mmx.hip -- measurement harness for the 1..8-token quantised mat-vec kernel (csrc/mmx_device.h) next to the round-2 kernel
// (k_mmt through mi_mmt_run, linked from the plugin's object files).  Not part of the product path.
//
//   build:  python scripts/build_lab.py            (-> gpurun_out/lab_mmx, travels to the GPU box with the snapshot)
//   run:    gpurun_out/lab_mmx <type q4_K|q6_K|q5_K|q8_0|q4_0> <rows> <k> <T> <plain|norm|swiglu|qkv> [stamps]
//
// Every timed sequence is NREP launches over NREP distinct weight tensors (> 300 MB in total: nothing is served from the Infinity
// Cache), each preceded by a small kernel that rewrites the activations from other CUs (as the producing launch of the real graph
// does: the consumer's first touch of x misses its XCD's L2).  Variants run interleaved in one process (A B C A B C ...); outputs are
// compared bit for bit with the round-2 kernel's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <string>
#include <algorithm>
#include <random>
#include "mmx_device.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

// what the plugin's objects expect from backend.cpp
void mi_allow_big_lds(const void * fn) { CK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024)); }
bool mi_ensure_tiled(ggml_tensor *) { return false; }

static uint16_t f2h(float f) { __half h = __float2half(f); uint16_t u; memcpy(&u, &h, 2); return u; }
static int type_of(const char * s) {
    if (!strcmp(s, "q4_K")) return GGML_TYPE_Q4_K; if (!strcmp(s, "q5_K")) return GGML_TYPE_Q5_K; if (!strcmp(s, "q6_K")) return GGML_TYPE_Q6_K;
    if (!strcmp(s, "q8_0")) return GGML_TYPE_Q8_0; if (!strcmp(s, "q4_0")) return GGML_TYPE_Q4_0;
    fprintf(stderr, "type?\n"); exit(1);
}
// random valid blocks, row-major (ggml layout): every byte random, the fp16 scale fields small finite numbers
static std::vector<uint8_t> random_weights(int type, int rows, int k, std::mt19937 & rng) {
    const auto tr = mi_traits(type);
    const size_t rb = mi_row_size(type, k), n = (size_t) rows * rb;
    std::vector<uint8_t> w(n);
    for (size_t i = 0; i < n; i += 4) { uint32_t v = rng(); memcpy(&w[i], &v, std::min<size_t>(4, n - i)); }
    std::uniform_real_distribution<float> ud(0.002f, 0.02f);
    const size_t nb = n / tr.size;
    for (size_t b = 0; b < nb; ++b) {
        uint8_t * p = &w[b * tr.size];
        auto put = [&](int off, float v) { uint16_t h = f2h(v); memcpy(p + off, &h, 2); };
        switch (type) {
            case GGML_TYPE_Q4_K: case GGML_TYPE_Q5_K: put(0, ud(rng)); put(2, ud(rng)); break;
            case GGML_TYPE_Q6_K: put(208, ud(rng)); break;
            case GGML_TYPE_Q8_0: case GGML_TYPE_Q4_0: put(0, ud(rng)); break;
        }
    }
    return w;
}

__global__ void k_rewrite(float * x, const float * src, int n, float eps) {        // x <- src * (1 + eps): a write from all over the chip
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) x[i] = src[i] * (1.0f + eps);
}

struct variant { std::string name; int pfd; int flags; bool old; };

template <int TYPE, bool DUAL, bool HOIST> static void launch_mmx(hipStream_t st, const mmvq_launch & L, int T, int want_depth, int flags, unsigned long long * stamps, bool stamp, const mx_next & nx) {
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int TILE = mq_tfrag<TYPE>::TILE;
    int nbuf = MX_F_NBUF(flags), batch = MX_F_BATCH(flags);
    static unsigned epoch = 0; flags |= (int)((++epoch & 0xfffu) << 20);
    while (batch > 1 && mx_ring_depth(T, L.k, DUAL, Q80, nbuf, TILE, 2, batch) < 2) { if (nbuf > 1) nbuf = 1; else --batch; }
    if (nbuf >= 2 && mx_ring_depth(T, L.k, DUAL, Q80, 2, TILE, 2, batch) < 2) nbuf = 1;
    flags = (flags & ~(3 | (7 << 16))) | nbuf | (batch << 16);
    const int depth = mx_ring_depth(T, L.k, DUAL, Q80, nbuf, TILE, want_depth, batch);
    if (depth < 1) { fprintf(stderr, "no LDS for a tile ring\n"); exit(1); }
    flags |= depth << 12;
    const size_t lds = mx_lds_bytes(T, L.k, DUAL, Q80, nbuf, batch) + (size_t) 16 * depth * TILE;
    int total = 0;
    if (DUAL) total = L.m[0].rows / 16; else for (int i = 0; i < L.n_mat; ++i) total += L.m[i].rows / 16;
    const int grid = total < 256 ? total : 256;
    auto fn = k_mmx<TYPE, DUAL, HOIST, true>;      // the lab runs the stamp instantiation only (stamps == nullptr: a scalar branch per stamp point)
    static bool once = false; if (!once) { mi_allow_big_lds((const void *) fn); once = true; }
    fn<<<grid, 1024, lds, st>>>(L, T, flags, stamp ? stamps : nullptr, (flags & 0x400) ? nx : mx_next{ nullptr, 0, 0 });
}
template <int TYPE> static void launch_type(hipStream_t st, const mmvq_launch & L, int T, int pfd, int flags, unsigned long long * stamps, bool stamp, const mx_next & nx) {
    const bool hoist = (flags & 0x200) && L.k <= 4096 && mq_aop<TYPE>::HAVE;
    if (L.swiglu) { if (hoist) launch_mmx<TYPE, true, true>(st, L, T, pfd, flags, stamps, stamp, nx); else launch_mmx<TYPE, true, false>(st, L, T, pfd, flags, stamps, stamp, nx); }
    else          { if (hoist) launch_mmx<TYPE, false, true>(st, L, T, pfd, flags, stamps, stamp, nx); else launch_mmx<TYPE, false, false>(st, L, T, pfd, flags, stamps, stamp, nx); }
}
static void launch_any(hipStream_t st, int type, const mmvq_launch & L, int T, int pfd, int flags, unsigned long long * stamps, bool stamp, const mx_next & nx) {
    switch (type) {
        case GGML_TYPE_Q4_K: launch_type<GGML_TYPE_Q4_K>(st, L, T, pfd, flags, stamps, stamp, nx); break;
        case GGML_TYPE_Q6_K: launch_type<GGML_TYPE_Q6_K>(st, L, T, pfd, flags, stamps, stamp, nx); break;
#ifdef LAB_ALL_TYPES
        case GGML_TYPE_Q5_K: launch_type<GGML_TYPE_Q5_K>(st, L, T, pfd, flags, stamps, stamp, nx); break;
        case GGML_TYPE_Q8_0: launch_type<GGML_TYPE_Q8_0>(st, L, T, pfd, flags, stamps, stamp, nx); break;
        case GGML_TYPE_Q4_0: launch_type<GGML_TYPE_Q4_0>(st, L, T, pfd, flags, stamps, stamp, nx); break;
#endif
        default: fprintf(stderr, "lab built without this type\n"); exit(1);
    }
}

int main(int argc, char ** argv) {
    if (argc < 6) { fprintf(stderr, "usage: lab_mmx type rows k T mode [stamps]\n"); return 1; }
    const int type = type_of(argv[1]), rows = atoi(argv[2]), k = atoi(argv[3]), T = atoi(argv[4]);
    const std::string mode = argv[5];
    const bool want_stamps = argc > 6 && !strcmp(argv[6], "stamps");
    const int nmat = mode == "swiglu" ? 2 : (mode == "qkv" ? 3 : 1);
    const bool norm = mode != "plain";
    std::mt19937 rng(1234);
    hipStream_t st; CK(hipStreamCreate(&st));

    const size_t rb = mi_row_size(type, k), wbytes = (size_t) rows * rb;
    const int nrep = std::max(4, (int)(400e6 / (double)(wbytes * nmat)) + 1);
    // one random matrix per member, tiled once, copied to every repetition's buffers
    std::vector<char *> W((size_t) nrep * nmat);
    for (int m = 0; m < nmat; ++m) {
        std::vector<uint8_t> h = random_weights(type, rows, k, rng);
        char * raw; CK(hipMalloc((void **) &raw, wbytes)); CK(hipMemcpy(raw, h.data(), wbytes, hipMemcpyHostToDevice));
        char * tiled; CK(hipMalloc((void **) &tiled, wbytes));
        mi_tile_permute(st, raw, tiled, type, rows, k, true); CK(hipStreamSynchronize(st));
        for (int r = 0; r < nrep; ++r) { char * d; CK(hipMalloc((void **) &d, wbytes + 4096)); CK(hipMemcpy(d, tiled, wbytes, hipMemcpyDeviceToDevice)); W[(size_t) r * nmat + m] = d; }
        CK(hipFree(raw)); CK(hipFree(tiled));
    }
    std::normal_distribution<float> nd(0.f, 1.f);
    std::vector<float> hx((size_t) T * k), hw(k), hres((size_t) T * rows);
    for (auto & v : hx) v = nd(rng);
    for (auto & v : hw) v = 1.0f + 0.1f * nd(rng);
    for (auto & v : hres) v = nd(rng);
    float * xsrc, * x, * nw, * res;
    CK(hipMalloc((void **) &xsrc, hx.size()*4)); CK(hipMalloc((void **) &x, hx.size()*4)); CK(hipMalloc((void **) &nw, hw.size()*4)); CK(hipMalloc((void **) &res, hres.size()*4));
    CK(hipMemcpy(xsrc, hx.data(), hx.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(x, hx.data(), hx.size()*4, hipMemcpyHostToDevice));
    CK(hipMemcpy(nw, hw.data(), hw.size()*4, hipMemcpyHostToDevice)); CK(hipMemcpy(res, hres.data(), hres.size()*4, hipMemcpyHostToDevice));
    const size_t obytes = (size_t) T * rows * 4 * (mode == "qkv" ? 3 : 1);
    float * out_ref, * out_new, * normout;
    CK(hipMalloc((void **) &out_ref, obytes)); CK(hipMalloc((void **) &out_new, obytes)); CK(hipMalloc((void **) &normout, (size_t) T*k*4));
    mi_act_cache cache; cache.slot_bytes = (size_t) 512 << 10; CK(hipMalloc((void **) &cache.pool, MI_ACT_SLOTS * cache.slot_bytes));
    unsigned long long * stamps; const size_t nst = (size_t) 256 * 16 * MX_NSTAMP; CK(hipMalloc((void **) &stamps, nst * 8));

    auto make = [&](int r, float * out) {
        mmvq_launch L{}; L.k = k; L.tiled = 1; L.n_mat = nmat; L.swiglu = mode == "swiglu";
        L.act.X = x; L.act.xs = k; L.act.norm = norm; L.act.norm_w = norm ? nw : nullptr; L.act.eps = 1e-6f;
        if (norm) { L.act.norm_out = normout; L.act.norm_os = k; }
        for (int m = 0; m < nmat; ++m) {
            mmvq_mat & M = L.m[m]; M.W = W[(size_t) r * nmat + m]; M.row_bytes = rb; M.rows = rows; M.epi = EPI_F32;
            M.out = (char *)(out + (mode == "qkv" ? (size_t) m * T * rows : 0)); M.o_row = 4; M.o_tok = (int64_t) rows * 4;
            M.res = (mode == "plain") ? res : nullptr; M.r_tok = rows;
        }
        return L;
    };

    std::vector<variant> vars;
    vars.push_back({ "r2 k_mmt", 0, 0, true });
    auto add = [&](int d, int pf, int hoist, int batch, int nxt) { vars.push_back({ "mmx d" + std::to_string(d) + " pf" + std::to_string(pf) + (hoist ? " hoist" : "") + " g" + std::to_string(batch) + (nxt ? " next" : ""), d, 2 | (pf << 4) | (hoist << 9) | (batch << 16) | (nxt << 10), false }); };
    add(2, 2, 1, 1, 0); add(2, 2, 1, 1, 1); add(4, 2, 1, 2, 0); add(4, 2, 1, 2, 1); add(4, 2, 1, 4, 1);
    auto run_one = [&](const variant & v, int r, float * out, bool stamp) {
        mmvq_launch L = make(r, out);
        if (v.old) mi_mmt_run(st, type, T, L, &cache, nullptr);
        else {
            // the next launch of the timed sequence streams repetition r + 1's first matrix: its row groups in tile order
            const mx_next nx = { W[(size_t)((r + 1) % nrep) * nmat], (int)(rb * 16), rows / 16 };
            launch_any(st, type, L, T, v.pfd, v.flags, stamps, stamp, nx);
        }
    };
    // ---- correctness: bit for bit against the round-2 kernel
    std::vector<float> ref(obytes / 4), got(obytes / 4), nref((size_t) T*k), ngot((size_t) T*k);
    CK(hipMemset(out_ref, 0xff, obytes)); CK(hipMemset(normout, 0, (size_t) T*k*4));
    run_one(vars[0], 0, out_ref, false); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(ref.data(), out_ref, obytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(nref.data(), normout, (size_t) T*k*4, hipMemcpyDeviceToHost));
    bool all_ok = true;
    for (size_t vi = 1; vi < vars.size(); ++vi) {
        CK(hipMemset(out_new, 0xff, obytes)); CK(hipMemset(normout, 0, (size_t) T*k*4));
        run_one(vars[vi], 0, out_new, false); CK(hipStreamSynchronize(st));
        CK(hipMemcpy(got.data(), out_new, obytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(ngot.data(), normout, (size_t) T*k*4, hipMemcpyDeviceToHost));
        // same arithmetic, but the compiler contracts a*b - c*d differently per instantiation: compare to a few ulp of the largest output
        size_t bad = 0; double maxd = 0, maxr = 0;
        for (size_t i = 0; i < ref.size(); ++i) maxr = std::max(maxr, (double) fabsf(ref[i]));
        for (size_t i = 0; i < got.size(); ++i) { const double d = fabs((double) got[i] - ref[i]); maxd = std::max(maxd, d); if (!(d <= 2e-6 * maxr)) ++bad; }
        size_t nbad = 0; if (norm) for (size_t i = 0; i < ngot.size(); ++i) if (memcmp(&ngot[i], &nref[i], 4)) ++nbad;
        if (bad || nbad) { all_ok = false; printf("MISMATCH %-24s: %zu of %zu outputs differ by more than 2e-6 of max |ref| %.3g (max abs %.3g), norm_out %zu differ\n", vars[vi].name.c_str(), bad, got.size(), maxr, maxd, nbad); }
        else if (vi == 1) printf("  max |new - r2| = %.3g at max |ref| = %.3g\n", maxd, maxr);
    }
    printf("%s rows %d k %d T %d %s: nrep %d, %.1f MB per launch; outputs %s\n", argv[1], rows, k, T, mode.c_str(), nrep, wbytes * nmat / 1e6, all_ok ? "equal to k_mmt within 2e-6 of max" : "DIFFER");

    // ---- timing: interleaved rounds, each NREP x (rewrite x ; mat-vec)
    const int rounds = 9;
    std::vector<std::vector<double>> us(vars.size());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // the rewrite launches alone (subtracted): same count, same stream
    std::vector<double> base_us;
    for (int it = 0; it < rounds + 1; ++it) {
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < nrep; ++r) k_rewrite<<<256, 256, 0, st>>>(x, xsrc, T*k, 1e-7f * r);
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) base_us.push_back(ms * 1e3 / nrep);
    }
    std::sort(base_us.begin(), base_us.end());
    for (int it = 0; it < rounds + 1; ++it) {
        for (size_t vi = 0; vi < vars.size(); ++vi) {
            CK(hipEventRecord(e0, st));
            for (int r = 0; r < nrep; ++r) { k_rewrite<<<256, 256, 0, st>>>(x, xsrc, T*k, 1e-7f * r); run_one(vars[vi], r, out_new, false); }
            CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it) us[vi].push_back(ms * 1e3 / nrep);
        }
    }
    const double b = base_us[base_us.size() / 2];
    printf("  rewrite-only launch: %.2f us (subtracted below)\n", b);
    for (size_t vi = 0; vi < vars.size(); ++vi) {
        std::sort(us[vi].begin(), us[vi].end());
        const double med = us[vi][us[vi].size() / 2] - b, mn = us[vi][0] - b;
        printf("  %-26s median %7.2f us  min %7.2f us   %6.0f GB/s\n", vars[vi].name.c_str(), med, mn, wbytes * nmat / med / 1e3);
    }
    // ---- phase stamps of one variant (diagnostic instantiation)
    if (want_stamps) {
        const char * sel = argc > 7 ? argv[7] : nullptr;
        for (size_t vi = 1; vi < vars.size(); ++vi) {
            if (sel ? vars[vi].name != sel : vi != 1) continue;
            std::vector<unsigned long long> hs(nst);
            for (int r = 0; r < 6; ++r) { k_rewrite<<<256, 256, 0, st>>>(x, xsrc, T*k, 1e-7f * r); CK(hipMemsetAsync(stamps, 0, nst * 8, st)); run_one(vars[vi], r, out_new, true); }
            CK(hipStreamSynchronize(st)); CK(hipMemcpy(hs.data(), stamps, nst * 8, hipMemcpyDeviceToHost));
            static const char * names[MX_NSTAMP] = { "entry", "image share done", "after prologue barrier", "first group units done", "after reduce barrier", "first epilogue done", "exit",
                                                     "activations landed", "partial sums parked", "after norm barrier", "norm scale known", "" };
            static const int order[] = { 0, 7, 8, 9, 10, 1, 2, 3, 4, 5, 6 };
            unsigned long long t0 = ~0ull;
            for (size_t i = 0; i < nst; i += MX_NSTAMP) if (hs[i] && hs[i] < t0) t0 = hs[i];
            printf("  stamps %s (us from the first wave's entry; min / median / max over waves):\n   ", vars[vi].name.c_str());
            for (int oi : order) {
                std::vector<double> v;
                for (size_t i = 0; i < nst; i += MX_NSTAMP) if (hs[i] && hs[i + oi]) v.push_back((hs[i + oi] - t0) / 100.0);
                if (v.empty()) continue;
                std::sort(v.begin(), v.end());
                printf(" %s %.2f/%.2f/%.2f |", names[oi], v.front(), v[v.size()/2], v.back());
            }
            printf("\n");
        }
    }
    return all_ok ? 0 : 2;
}
