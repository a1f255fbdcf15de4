// kernels_attn.hip -- the attention sub-graph of llm_build_kqv without flash attention
// (R/src/llama.cpp:706-828) as ONE kernel for small token batches (draft steps, tree verification):
//     kq  = MUL_MAT(K f16 [d, n_kv, H_kv], q f32 [d, T, H])
//     p   = SOFT_MAX(kq * scale + mask)            mask f32 [n_kv, pad64(T)]: the tree / causal mask with -INF
//     kqv = MUL_MAT(V f16 [n_kv, d, H_kv] (transposed cache), p)
//     out = CONT(PERMUTE(kqv))  -> [d*H, T]
// replacing 2 mat-mul launches (R/ggml/src/ggml-cuda/mmv.cu:197 or batched hipBLAS ggml-cuda.cu:1693), soft_max_f32
// (softmax.cu:23) and a cpy.  Numerics follow the CPU backend: q and p are rounded to f16 before the dot products
// with the f16 cache (vec_dot_type of F16, R/ggml/src/ggml-cpu/ggml-cpu.c:260-264), products accumulate in fp32,
// the soft-max denominator in double (ggml-cpu.c:9122-9129).
//
// One 256-thread block per (head, tile of <= 8 or 16 tokens, slice of the output dims).  Both products run on the
// matrix cores (v_mfma_f32_16x16x32_f16: f16 x f16 products are exact in fp32, accumulation in fp32 like the CPU
// backend's vec_dot_f16): a 16-byte load IS one lane's A fragment (K rows: 8 consecutive head dims of one cell; V^T
// rows: 8 consecutive cells of one head dim), the B fragment is the f16-rounded q / p row of token lane%16.  Scores
// live in LDS (fp32), the f16 probabilities in a second LDS image; the soft-max keeps the CPU's double sum.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <hip/hip_fp16.h>
#include "kernels.h"
#include "mi355x_common.h"
#include "lane_ops.h"
#include <mutex>
#include <cstdlib>

#define WAVE 64
typedef int      i32x4 __attribute__((ext_vector_type(4)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 as_h8(const i32x4 v) { return __builtin_bit_cast(f16x8, v); }
__device__ __forceinline__ float rnd16(float v) { return __half2float(__float2half_rn(v)); }

// LDS image: sc [tt][n_kv + 4] fp32 scores -- the f16 probabilities of a token overwrite the front half of its own score row (probability i
// lands on bytes of score i/2, which the same wave read in an earlier or the same step) --, red [4][64] float4 partial tiles.  32 bytes
// per cell at 8 tokens: the fused kernel reaches n_kv = 4096 (Llama-2's context) inside 150 KiB; a separate probability image stopped at ~3000
static inline size_t attn_lds_bytes(int n_kv, int tt) { return (size_t) tt * (n_kv + 4) * 4 + 4 * 64 * 16 + 4 * 4 * 1024; }    // + per-wave V staging (row-major V only)

// diagnostic stamps (GGML_MI355X_ATTN_STAMPS, scripts/attn_stamps.py): the STAMP = true instantiation writes s_memrealtime at the phase
// boundaries into a debug buffer nothing else reads; the product instantiation contains no stamp code
__device__ unsigned long long * g_attn_stamps = nullptr;
template <bool STAMP> __device__ __forceinline__ void attn_stamp(int idx) {
    if constexpr (STAMP) {
        if ((threadIdx.x & 63) == 0 && g_attn_stamps) {
            const int blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
            if (blk < 512) g_attn_stamps[((size_t) blk * 4 + (threadIdx.x >> 6)) * 8 + idx] = __builtin_amdgcn_s_memrealtime();
        }
    }
}
template <int D, bool STAMP = false> __global__ void __launch_bounds__(256) k_attn_small(const mi_attn_args a, const int tt) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    attn_stamp<STAMP>(0);
    const int n_kv = a.n_kv;
    const int ldS = n_kv + 4, ldP = 2*ldS;                  // ldP: the same row pitch counted in f16 elements
    float    * sc  = (float *) lds;
    _Float16 * ph  = (_Float16 *) lds;                     // probabilities in place (see attn_lds_bytes)
    f32x4    * red = (f32x4 *)(lds + (size_t) tt * ldS * 4);
    _Float16 * stg = (_Float16 *)((char *) red + 4*64*16);              // [4 waves][4 chunks][32 cells][16 head dims]: row-major V -> B operands
    const int h = blockIdx.x, t0 = blockIdx.y * tt;
    const int nt = min(tt, a.T - t0);
    const int hk = h / (a.H / a.H_kv);
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int col = lane & 15, grp = lane >> 4;            // MFMA maps: A row / B column = lane&15, k = 8*(lane>>4) + j
    const char * kb = (const char *) a.k + (int64_t) hk * a.k_nb2;
    const char * vb = (const char *) a.v + (int64_t) hk * a.v_nb2;
    constexpr int NS = D / 32;                             // MFMA k-steps over the head dim

    // phase 3 geometry, known up front: blockIdx.z takes TPB of the D/16 output tiles; the 4 waves split (tile, cell range)
    const int tpb = (D/16) / gridDim.z;                    // 1, 2 or 4 tiles per block
    const int wpt = 4 / tpb;                               // waves sharing one tile
    const int tile = blockIdx.z * tpb + wave / wpt, part = wave % wpt;
    const int dd0 = tile * 16;
    constexpr int VU = 4;                                  // 32-cell chunks in flight
    // the first V^T fragments do not depend on the soft-max: their loads go out now and land during phases 1 and 2
    i32x4 vf0[VU];
#pragma unroll
    for (int u = 0; u < VU; ++u) {
        const int i = part*32 + u*wpt*32 + 8*grp;
        if (a.v_row) { const int cell = part*32 + u*wpt*32 + (lane >> 1); vf0[u] = (cell < n_kv) ? *(const i32x4 *)(vb + (int64_t) cell * a.v_nb1 + (int64_t)(dd0 + 8*(lane & 1)) * 2) : (i32x4)(0); }
        else vf0[u] = (i < n_kv) ? *(const i32x4 *)(vb + (int64_t)(dd0 + col) * a.v_nb1 + (int64_t) i * 2) : (i32x4)(0);
    }
    // ---- phase 1: scores[cell][t] = K[cell][:] . q[t][:]      (A = 16 cells x 32 dims, B = 32 dims x 16 tokens)
    // Every global load of the phase -- the first K tiles, their mask values, the q rows -- is requested before anything is waited
    // for (the kernel is a chain of latencies, not of bytes: serialising q -> K cost a full memory round trip).
    constexpr int CU = 4;                                  // cell tiles in flight per wave (CU*NS 16-byte loads per lane): n_kv <= 256 is one round trip
    i32x4 kf[CU][NS]; f32x4 mk[CU];
    auto load_k = [&](int c0) {
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int cell = c0 + u*64 + col;
#pragma unroll
            for (int s = 0; s < NS; ++s) kf[u][s] = (cell < n_kv) ? *(const i32x4 *)(kb + (int64_t) cell * a.k_nb1 + (32*s + 8*grp) * 2) : (i32x4)(0);
        }
        // the mask values of the C elements this lane will own go out with the K loads (phase 2 then never touches global memory)
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int r0 = c0 + u*64 + 4*grp;
            mk[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (a.mask && col < nt && r0 < n_kv) {
                const char * mrow = (const char *) a.mask + (int64_t)(t0 + col) * a.mask_nb1;
                if (a.mask_f16) { const __half2 * m = (const __half2 *)(mrow + (int64_t) r0 * 2); const float2 lo = __half22float2(m[0]), hi = __half22float2(m[1]); mk[u] = (f32x4){lo.x, lo.y, hi.x, hi.y}; }
                else mk[u] = *(const f32x4 *)(mrow + (int64_t) r0 * 4);
            }
        }
    };
    if (wave*16 < n_kv) load_k(wave*16);
    f32x4 qr[NS][2];
    const bool q16 = (((uintptr_t) a.q | (uintptr_t) a.q_nb1 | (uintptr_t) a.q_nb2) & 15) == 0;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float * qp = (const float *)((const char *) a.q + (int64_t)(t0 + (col < nt ? col : 0)) * a.q_nb1 + (int64_t) h * a.q_nb2) + 32*s + 8*grp;
        if (q16) { qr[s][0] = *(const f32x4 *) qp; qr[s][1] = *(const f32x4 *)(qp + 4); }      // the usual case: two 16-byte loads
        else { __builtin_memcpy(&qr[s][0], qp, 16); __builtin_memcpy(&qr[s][1], qp + 4, 16); }   // rows are only known to be 4-byte aligned
    }
    attn_stamp<STAMP>(1);                                  // everything requested
    f16x8 qf[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qf[s][j] = col < nt ? (_Float16) qr[s][j >> 2][j & 3] : (_Float16) 0.f;
    }
    for (int c0 = wave*16; c0 < n_kv; c0 += 4*16*CU) {
        if (c0 != wave*16) load_k(c0);
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < NS; ++s) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(kf[u][s]), qf[s], c, 0, 0, 0);
            const int r0 = c0 + u*64 + 4*grp;              // C: column (token) = lane&15, rows (cells) = 4*(lane>>4) + 0..3
            if (col < nt && r0 < n_kv) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { c[r] = __fmul_rn(c[r], a.scale); if (a.mask) c[r] = __fadd_rn(c[r], mk[u][r]); }     // scale, then mask: the order of ggml's soft_max
                *(f32x4 *)(sc + col*ldS + r0) = c;
            }
        }
    }
    attn_stamp<STAMP>(2);
    __syncthreads();
    attn_stamp<STAMP>(3);
    // ---- phase 2: soft-max, 256 / tt lanes per token row (32 for draft / verification batches), probabilities rounded to f16 (the
    // vec_dot_type of the f16 V cache)
    {
        const int lg = 256 / tt;                            // 32 or 16: a power of two inside one wave
        const int t = threadIdx.x / lg, sub = threadIdx.x % lg;
        if (t < nt) {
            float * row = sc + t*ldS;
            _Float16 * prow = ph + t*ldP;
            // one LDS read and one write per score instead of three round trips (max, exp, normalise passes over the LDS row): the
            // phase is a latency chain at one wave per SIMD.  Same per-lane order of operations as the loop form below.  Scores per lane
            // kept in registers: 12 at 32 lanes per token, 24 at 16 lanes per token (16-token tiles of big batches): n_kv <= 384 either way
            auto in_regs = [&](auto nr_tag) {
                constexpr int NR = decltype(nr_tag)::value;
                float v[NR];
#pragma unroll
                for (int j = 0; j < NR; ++j) { const int i = sub + j*lg; v[j] = i < n_kv ? row[i] : -INFINITY; }
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < NR; ++j) mx = fmaxf(mx, v[j]);
                mx = row_max_f(mx); if (lg == 32) mx = max_xw<16>(mx);
                double sum = 0.0;
#pragma unroll
                for (int j = 0; j < NR; ++j) { if (sub + j*lg < n_kv) { const float e = (v[j] == -INFINITY) ? 0.0f : expf(v[j] - mx); v[j] = e; sum += (double) e; } }
                sum = row_sum_d(sum); if (lg == 32) sum = sum_xw<16>(sum);
                const float inv = (float)(1.0 / sum);
#pragma unroll
                for (int j = 0; j < NR; ++j) { const int i = sub + j*lg; if (i < n_kv) prow[i] = (_Float16)(v[j] * inv); }
            };
            if (lg == 32 && n_kv <= 12*32) in_regs(std::integral_constant<int, 12>{});
            else if (lg == 16 && n_kv <= 24*16) in_regs(std::integral_constant<int, 24>{});
            else {
            float mx = -INFINITY;
            for (int i = sub; i < n_kv; i += lg) mx = fmaxf(mx, row[i]);
            mx = row_max_f(mx); if (lg == 32) mx = max_xw<16>(mx);            // 16 or 32 lanes per token: DPP row steps + one VALU lane-pair step
            double sum = 0.0;
            for (int i = sub; i < n_kv; i += lg) { const float e = (row[i] == -INFINITY) ? 0.0f : expf(row[i] - mx); row[i] = e; sum += (double) e; }
            sum = row_sum_d(sum); if (lg == 32) sum = sum_xw<16>(sum);
            const float inv = (float)(1.0 / sum);
            for (int i = sub; i < n_kv; i += lg) prow[i] = (_Float16)(row[i] * inv);
            }
        }
    }
    attn_stamp<STAMP>(4);
    __syncthreads();
    attn_stamp<STAMP>(5);
    // ---- phase 3: out[t][dd] = sum_i V^T[dd][i] * p[t][i]      (A = 16 head dims x 32 cells, B = 32 cells x 16 tokens)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!a.v_row) {
    for (int i0 = part*32; i0 < n_kv; i0 += wpt*32*VU) {
        i32x4 vf[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int i = i0 + u*wpt*32 + 8*grp;
            if (i0 == part*32) vf[u] = vf0[u];
            else vf[u] = (i < n_kv) ? *(const i32x4 *)(vb + (int64_t)(dd0 + col) * a.v_nb1 + (int64_t) i * 2) : (i32x4)(0);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int i = i0 + u*wpt*32 + 8*grp;
            const i32x4 pf = (col < nt && i < n_kv) ? *(const i32x4 *)(ph + col*ldP + i) : (i32x4)(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(vf[u]), as_h8(pf), acc, 0, 0, 0);
        }
    }
    } else {
    // row-major V (FLASH_ATTN_EXT): out[t][dd] = sum_i p[t][i] * V[i][dd] with A = p (16 tokens x 32 cells) and B = V (32 cells x 16 head dims).
    // A lane's B fragment is a COLUMN of V (8 cells of one head dim): the 32 x 16 chunk is fetched with one coalesced 16-byte load per lane
    // (cell = lane/2, 8 head dims each), parked in a wave-private LDS tile, and read back as 8 halves.
    _Float16 * my = stg + (size_t) wave * 4 * 512;
    for (int i0 = part*32; i0 < n_kv; i0 += wpt*32*VU) {
        i32x4 vr[VU];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int cell = i0 + u*wpt*32 + (lane >> 1);
            if (i0 == part*32) vr[u] = vf0[u];
            else vr[u] = (cell < n_kv) ? *(const i32x4 *)(vb + (int64_t) cell * a.v_nb1 + (int64_t)(dd0 + 8*(lane & 1)) * 2) : (i32x4)(0);
        }
#pragma unroll
        for (int u = 0; u < VU; ++u) *(i32x4 *)(my + u*512 + (lane >> 1)*16 + 8*(lane & 1)) = vr[u];
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            const int i = i0 + u*wpt*32 + 8*grp;
            f16x8 bv;
#pragma unroll
            for (int j = 0; j < 8; ++j) bv[j] = my[u*512 + (8*grp + j)*16 + col];
            const i32x4 pf = (col < nt && i < n_kv) ? *(const i32x4 *)(ph + col*ldP + i) : (i32x4)(0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_h8(pf), bv, acc, 0, 0, 0);       // C: column = head dim lane&15, rows = tokens 4*(lane>>4) + 0..3
        }
    }
    }
    attn_stamp<STAMP>(6);
    if (wpt > 1) {
        if (part) red[wave*64 + lane] = acc;
        __syncthreads();
        if (part == 0) for (int w = 1; w < wpt; ++w) acc += red[(wave + w)*64 + lane];
    }
    if (part == 0 && !a.v_row && col < nt) {               // C: column (token) = lane&15, rows (head dims) = 4*(lane>>4) + 0..3
        float * o = (float *)((char *) a.out + (int64_t)(dd0 + 4*grp)*4 + (int64_t) h * a.o_nb1 + (int64_t)(t0 + col) * a.o_nb2);
        o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
    }
    if (part == 0 && a.v_row) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int tok = 4*grp + r; if (tok < nt) *(float *)((char *) a.out + (int64_t)(dd0 + col)*4 + (int64_t) h * a.o_nb1 + (int64_t)(t0 + tok) * a.o_nb2) = acc[r]; }
    }
    attn_stamp<STAMP>(7);
}
#ifdef MI_LAB          // the STAMP = true instantiation and its plumbing exist in the lab build only (build.py --lab)
static unsigned long long * g_attn_stamp_dev = nullptr;
static bool attn_stamps_on() {
    static const bool on = [] {
        if (!mi_lab_env("GGML_MI355X_ATTN_STAMPS")) return false;
        const size_t n = (size_t) 512 * 4 * 8 * 8;
        HIP_CHECK(hipMalloc((void **) &g_attn_stamp_dev, n)); HIP_CHECK(hipMemset(g_attn_stamp_dev, 0, n));
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_attn_stamps), &g_attn_stamp_dev, sizeof(void *)));
        return true;
    }();
    return on;
}
extern "C" __attribute__((visibility("default"))) int ggml_backend_mi355x_attn_stamps(unsigned long long * out) {
    if (!g_attn_stamp_dev) return 0;
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, g_attn_stamp_dev, (size_t) 512 * 4 * 8 * 8, hipMemcpyDeviceToHost));
    return 512 * 4 * 8;
}
#endif

static int attn_tokens_per_block(const mi_attn_args & a) { return (a.T > 8 && attn_lds_bytes(a.n_kv, 16) <= 150*1024) ? 16 : 8; }

bool mi_attn_small_supported(const mi_attn_args & a) {
    if (!(a.d == 64 || a.d == 128)) return false;
    if (a.n_kv % 8 || a.n_kv <= 0 || attn_lds_bytes(a.n_kv, 8) > 150*1024) return false;
    if (a.H % a.H_kv) return false;
    if (((uintptr_t) a.k | (uintptr_t) a.v | (uintptr_t) a.k_nb1 | (uintptr_t) a.k_nb2 | (uintptr_t) a.v_nb1 | (uintptr_t) a.v_nb2) & 15) return false;
    if (((uintptr_t) a.q | (uintptr_t) a.q_nb1 | (uintptr_t) a.q_nb2) & 3) return false;
    if (a.mask && (((uintptr_t) a.mask | (uintptr_t) a.mask_nb1) & (a.mask_f16 ? 7 : 15))) return false;     // 4 mask values per lane in one load
    return true;
}

void mi_op_attn_small(hipStream_t st, const mi_attn_args & a) {
    const int tt = attn_tokens_per_block(a);
    const size_t lds = attn_lds_bytes(a.n_kv, tt);
    const int tiles = (a.T + tt - 1) / tt;
    const int ntile = a.d / 16;                                   // output tiles of 16 head dims
    // more blocks when heads x token tiles under-fill the chip; a block keeps 1, 2 or 4 output tiles
    // (160 (head, tile) pairs of a 69-token verification batch: 2 x 160 blocks run 7.7 ms per round, 4 x 160 blocks 8.0, 8 x 160 blocks 8.4 -- every
    //  extra split recomputes K.q and the soft-max of its tile: profiles/r03_attn_dsplit_ab.txt)
    int dsplit = (a.H * tiles >= 128) ? ntile/4 : ntile;
    if (dsplit < 1) dsplit = 1;
    { static const int force = [] { const char * e = mi_lab_env("GGML_MI355X_ATTN_DSPLIT"); return e ? atoi(e) : 0; }(); if (force >= ntile/4 && force <= ntile && a.T > 8 && ntile % force == 0) dsplit = force; }      // lab A/B (big batches)
    { static const int force = [] { const char * e = mi_lab_env("GGML_MI355X_ATTN_DSPLIT_SMALL"); return e ? atoi(e) : 0; }(); if (force >= ntile/4 && force <= ntile && a.T <= 8 && ntile % force == 0) dsplit = force; }      // lab A/B (<= 8 tokens)
    const dim3 grid(a.H, tiles, dsplit);
#ifdef MI_LAB
    if (a.d == 128 && attn_stamps_on()) {
        mi_allow_big_lds((const void *) k_attn_small<128, true>);
        k_attn_small<128, true><<<grid, 256, lds, st>>>(a, tt);
        return;
    }
#endif
    mi_allow_big_lds(a.d == 128 ? (const void *) k_attn_small<128> : (const void *) k_attn_small<64>);
    if (a.d == 128) k_attn_small<128><<<grid, 256, lds, st>>>(a, tt);
    else            k_attn_small<64><<<grid, 256, lds, st>>>(a, tt);
}
