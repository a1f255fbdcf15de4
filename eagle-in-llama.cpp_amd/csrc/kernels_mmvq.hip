// kernels_mmvq.hip -- quantised weight x f32 activation products for 1..8 tokens ("mat-vec"),
// the HBM-bound core of the EAGLE draft step and of tree verification.
//
// What it replaces in the reference:  quantize_q8_1 (R/ggml/src/ggml-cuda/quantize.cu:4) +
// mul_mat_vec_q (R/ggml/src/ggml-cuda/mmvq.cu:55) + vec_dot_*_q8_1 (vecdotq.cuh) -- 32-lane warps,
// 4-byte loads, Q8_1 activations.  What it computes is the CPU backend's arithmetic instead
// (that is the parity target): activations are quantised exactly as
//   quantize_row_q8_K_ref  (R/ggml/src/ggml-quants.c:2479-2512)  for Q4_K / Q5_K / Q6_K weights
//   quantize_row_q8_0_ref  (R/ggml/src/ggml-quants.c:194-215)    for Q4_0 / Q8_0 weights
// and the integer dot products are those of ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K / {q4_0,q8_0}_q8_0
// (R/ggml/src/ggml-cpu/ggml-cpu-quants.c scalar branches :7020-7078, q5_K/q6_K `#else` tails, :2592-2607),
// so every int32 partial sum is identical to the CPU's; only the order of the final fp32 adds differs.
//
// This file holds (1) k_mmvq, the dp4a kernel that serves single-token products (and Q4_0 up to 8 tokens), (2) the activation
// quantisers (k_quant_q8K / k_quant_act) that write the int8 image for the matrix-core kernel (kernels_mmq.hip) and for big
// T*k, and (3) mi_mmvq_run, the dispatcher between the two kernels (token passes, k-chunks, the scratch-slot cache).
//
// k_mmvq on gfx950:
//   * prologue: with T*k <= 8192 the block quantises the activation columns straight into LDS (int8 + scales + bsums, RMS norm
//     folded in): nothing goes to HBM and the separate quantize launch of the reference disappears; above that it copies the
//     image a quantiser launch left in an HBM scratch slot;
//   * main loop: a wave owns R=2 weight rows at a time; 8 lanes cover one 256-element super-block with
//     one 16-byte load each (plus the shared 16-byte header), i.e. a wave instruction streams 8
//     super-blocks = 1152 B of Q4_K contiguous per row; loads are issued for both rows and two
//     k-steps before the first use so >=8 x 16 B per lane are in flight;
//   * v_dot4_i32_i8 on packed nibbles, per-lane fp32 partials, DPP reductions (no ds_bpermute);
//   * epilogues: residual / bias (+RELU), RoPE, fp16 cache stores, SwiGLU.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "kernels.h"
#include <algorithm>
#include <mutex>
#include <unordered_map>
#include <unordered_set>
#include <cstdlib>
#include <cstdio>
#include <vector>

#include "mmvq_device.h"


// ---------------------------------------------------------------- per-type weight fragments
// A fragment is what one lane owns of one weight row for one k-step: a 16-byte slice of quants plus the scales that
// go with it.  load() issues the global loads (nothing else), decode() unpacks nibbles and scales ONCE, mac() is the
// per-token work: integer dot products against the int8 activations the caller read from LDS (shared by all the
// fragments of the step), two int->float conversions and two FMAs.  Lanes past the end of a ragged last k-step keep
// d = 0 and a clamped (valid) address instead of branching.
struct act_regs { i32x4 lo, hi; int b0, b1; float dy; };

template <int TYPE> struct wfrag;

// Q4_K: lane (sb, c): c in 0..7 -> quant bytes [16c,16c+16): group g=c/2 (sub-blocks 2g | 2g+1 in low | high nibbles), half h=c%2
template <> struct wfrag<GGML_TYPE_Q4_K> {
    i32x4 hdr, qs; int sc0, sc1, m0, m1; float dw, mw;
    static __device__ __forceinline__ int nunits(int k) { return k / 256; }                      // units of a row: super-blocks
    static __device__ __forceinline__ bool valid(int unit, int lane, int k) { return unit < k / 256; }
    static __device__ __forceinline__ void act_off(int lane, int & sub, int & alo, int & ahi, int & blo, int & bhi) {
        const int c = lane & 7, g = c >> 1, h = c & 1; sub = lane >> 3; alo = 64*g + 16*h; ahi = alo + 32; blo = 4*g + h; bhi = blo + 2;
    }
    __device__ __forceinline__ void load(const char * row, int unit, int lane, int) { const char * b = row + unit*144; hdr = ld16(b); qs = ld16(b + 16 + 16*(lane & 7)); }
    __device__ __forceinline__ void decode(int lane, bool valid) {
        const int g = (lane & 7) >> 1;
        const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;                                           // scales[0..3], [4..7], [8..11]
        // get_scale_min_k4 (R/ggml/src/ggml-quants.c:631-638) for all eight sub-blocks at once, as the CPU's utmp/kmask trick does
        const uint32_t s_lo = u0 & 0x3f3f3f3fu, s_hi = (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u);
        const uint32_t m_lo = u1 & 0x3f3f3f3fu, m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u);
        const uint32_t sp = ((g < 2 ? s_lo : s_hi) >> (16*(g & 1))) & 0xffffu, mp = ((g < 2 ? m_lo : m_hi) >> (16*(g & 1))) & 0xffffu;
        sc0 = sp & 0xff; sc1 = sp >> 8; m0 = mp & 0xff; m1 = mp >> 8;
        dw = valid ? h2f((uint16_t)(hdr.x & 0xffff)) : 0.f; mw = valid ? h2f((uint16_t)((uint32_t) hdr.x >> 16)) : 0.f;
        const i32x4 q = qs; qs = q & 0x0F0F0F0F; hdr = (q >> 4) & 0x0F0F0F0F;                         // qs = low nibbles, hdr = high nibbles from here on
    }
    __device__ __forceinline__ float mac(const act_regs & a) const {
        const int si = __mul24(sc0, dot16(qs, a.lo)) + __mul24(sc1, dot16(hdr, a.hi));   // 6-bit x 23-bit: full-rate 24-bit multiply
        const int mi = __mul24(m0, a.b0) + __mul24(m1, a.b1);
        return (dw*a.dy)*(float) si - (mw*a.dy)*(float) mi;
    }
};
// Q5_K: as Q4_K plus one high bit per element from qh[32]
template <> struct wfrag<GGML_TYPE_Q5_K> {
    i32x4 hdr, qh, qs; int sc0, sc1, m0, m1; float dw, mw;
    static __device__ __forceinline__ int nunits(int k) { return k / 256; }
    static __device__ __forceinline__ bool valid(int unit, int lane, int k) { return unit < k / 256; }
    static __device__ __forceinline__ void act_off(int lane, int & sub, int & alo, int & ahi, int & blo, int & bhi) {
        const int c = lane & 7, g = c >> 1, h = c & 1; sub = lane >> 3; alo = 64*g + 16*h; ahi = alo + 32; blo = 4*g + h; bhi = blo + 2;
    }
    __device__ __forceinline__ void load(const char * row, int unit, int lane, int) { const char * b = row + unit*176; const int c = lane & 7; hdr = ld16(b); qh = ld16(b + 16 + 16*(c & 1)); qs = ld16(b + 48 + 16*c); }
    __device__ __forceinline__ void decode(int lane, bool valid) {
        const int g = (lane & 7) >> 1;
        const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;
        const uint32_t s_lo = u0 & 0x3f3f3f3fu, s_hi = (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u);
        const uint32_t m_lo = u1 & 0x3f3f3f3fu, m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u);
        const uint32_t sp = ((g < 2 ? s_lo : s_hi) >> (16*(g & 1))) & 0xffffu, mp = ((g < 2 ? m_lo : m_hi) >> (16*(g & 1))) & 0xffffu;
        sc0 = sp & 0xff; sc1 = sp >> 8; m0 = mp & 0xff; m1 = mp >> 8;
        dw = valid ? h2f((uint16_t)(hdr.x & 0xffff)) : 0.f; mw = valid ? h2f((uint16_t)((uint32_t) hdr.x >> 16)) : 0.f;
        const i32x4 q = qs, hb = qh;
        qs  = (q & 0x0F0F0F0F)        | (((hb >> (2*g))     & 0x01010101) << 4);
        hdr = ((q >> 4) & 0x0F0F0F0F) | (((hb >> (2*g + 1)) & 0x01010101) << 4);
    }
    __device__ __forceinline__ float mac(const act_regs & a) const {
        const int si = __mul24(sc0, dot16(qs, a.lo)) + __mul24(sc1, dot16(hdr, a.hi));   // 6-bit x 23-bit: full-rate 24-bit multiply
        const int mi = __mul24(m0, a.b0) + __mul24(m1, a.b1);
        return (dw*a.dy)*(float) si - (mw*a.dy)*(float) mi;
    }
};
// Q6_K: lane (sb, c): half n=c/4, cc=c%4: ql bytes [64n+16cc, +16): low nibbles -> elements 128n+16cc.., high -> +64;
// the two high bits come from qh[32n + 16(cc%2) ..] at bit offsets 2(cc/2) and 2(cc/2)+4; value - 32 (folded in through bsums).
template <> struct wfrag<GGML_TYPE_Q6_K> {
    i32x4 ql, qh; int sc0, sc1; float dw;
    static __device__ __forceinline__ int nunits(int k) { return k / 256; }
    static __device__ __forceinline__ bool valid(int unit, int lane, int k) { return unit < k / 256; }
    static __device__ __forceinline__ void act_off(int lane, int & sub, int & alo, int & ahi, int & blo, int & bhi) {
        const int c = lane & 7, n = c >> 2, cc = c & 3; sub = lane >> 3; alo = 128*n + 16*cc; ahi = alo + 64; blo = 8*n + cc; bhi = blo + 4;
    }
    __device__ __forceinline__ void load(const char * row, int unit, int lane, int) {
        const char * b = row + unit*210; const int c = lane & 7, n = c >> 2, cc = c & 3;
        ql = ld16(b + 64*n + 16*cc); qh = ld16(b + 128 + 32*n + 16*(cc & 1));
        const int8_t * s = (const int8_t *)(b + 192 + 8*n + cc);
        sc0 = s[0]; sc1 = s[4];
        uint16_t dh; __builtin_memcpy(&dh, b + 208, 2); dw = h2f(dh);
    }
    __device__ __forceinline__ void decode(int lane, bool valid) {
        const int sh = 2*((lane & 3) >> 1);
        const i32x4 l = ql, hb = qh;
        ql = (l & 0x0F0F0F0F)        | (((hb >> sh)       & 0x03030303) << 4);
        qh = ((l >> 4) & 0x0F0F0F0F) | (((hb >> (sh + 4)) & 0x03030303) << 4);
        if (!valid) dw = 0.f;
    }
    __device__ __forceinline__ float mac(const act_regs & a) const {
        const int si = __mul24(sc0, dot16(ql, a.lo) - 32*a.b0) + __mul24(sc1, dot16(qh, a.hi) - 32*a.b1);
        return (dw*a.dy)*(float) si;
    }
};
// Q8_0: two lanes per 32-element block (16 int8 each); the block's int32 sum is completed across the lane pair
// before the single fp32 multiply, as in ggml_vec_dot_q8_0_q8_0; only the even lane accumulates.
template <> struct wfrag<GGML_TYPE_Q8_0> {
    i32x4 qs; float dw;
    static __device__ __forceinline__ int nunits(int k) { return (k/32 + 3) / 4; }                 // unit = 4 blocks: 8 lanes
    static __device__ __forceinline__ bool valid(int unit, int lane, int k) { return unit*4 + ((lane & 7) >> 1) < k/32; }
    static __device__ __forceinline__ void act_off(int lane, int & sub, int & alo, int & ahi, int & blo, int & bhi) {
        const int c = lane & 7; sub = lane >> 3; alo = 32*(c >> 1) + 16*(c & 1); ahi = alo; blo = c >> 1; bhi = 0;
    }
    __device__ __forceinline__ void load(const char * row, int unit, int lane, int k) {
        const int c = lane & 7; const char * b = row + min(unit*4 + (c >> 1), k/32 - 1)*34; uint16_t dh; __builtin_memcpy(&dh, b, 2); dw = h2f(dh); qs = ld16(b + 2 + 16*(c & 1));
    }
    __device__ __forceinline__ void decode(int lane, bool valid) { if (!valid || (lane & 1)) dw = 0.f; }
    __device__ __forceinline__ float mac(const act_regs & a) const {
        int s = dot16(qs, a.lo);
        s += dpp_i<DPP_XOR1>(s);
        return (float) s * (dw*a.dy);
    }
};
// Q4_0: one lane per block: low nibbles -> elements 0..15, high -> 16..31, value - 8.
template <> struct wfrag<GGML_TYPE_Q4_0> {
    i32x4 qs, hi; float dw;
    static __device__ __forceinline__ int nunits(int k) { return (k/32 + 7) / 8; }                 // unit = 8 blocks: 8 lanes
    static __device__ __forceinline__ bool valid(int unit, int lane, int k) { return unit*8 + (lane & 7) < k/32; }
    static __device__ __forceinline__ void act_off(int lane, int & sub, int & alo, int & ahi, int & blo, int & bhi) {
        const int c = lane & 7; sub = lane >> 3; alo = 32*c; ahi = alo + 16; blo = c; bhi = 0;
    }
    __device__ __forceinline__ void load(const char * row, int unit, int lane, int k) {
        const char * b = row + min(unit*8 + (lane & 7), k/32 - 1)*18; uint16_t dh; __builtin_memcpy(&dh, b, 2); dw = h2f(dh); qs = ld16(b + 2);
    }
    __device__ __forceinline__ void decode(int lane, bool valid) { const i32x4 q = qs; qs = q & 0x0F0F0F0F; hi = (q >> 4) & 0x0F0F0F0F; if (!valid) dw = 0.f; }
    __device__ __forceinline__ float mac(const act_regs & a) const {
        const i32x4 ones = (i32x4)(0x01010101);
        const int s = dot16(qs, a.lo) + dot16(hi, a.hi) - 8*(dot16(ones, a.lo) + dot16(ones, a.hi));
        return ((float) s * dw) * a.dy;                                                              // (sumi*dx)*dy, ggml-cpu-quants.c:2606
    }
};
template <int TYPE> struct act_kind { static constexpr bool K = (TYPE == GGML_TYPE_Q4_K || TYPE == GGML_TYPE_Q5_K || TYPE == GGML_TYPE_Q6_K); };

// ---------------------------------------------------------------- the kernel
// One launch = up to three weight matrices of ONE quant type sharing the same activations (wq|wk|wv, or gate|up),
// each with its own epilogue:
//   EPI_F32       dst f32 (+ residual row: fused GGML_OP_ADD)
//   EPI_ROPE_F32  RoPE (mode NORM) on row pairs, f32 out                      (Qcur)
//   EPI_ROPE_F16  RoPE, then f16 store straight into the K cache view          (Kcur -> cpy -> k_cache_view)
//   EPI_F16       f16 store with row/token strides: the transposed V cache     (Vcur -> transpose -> cpy)
// DUAL: matrices 0/1 are ffn_gate/ffn_up, a wave computes the same rows of both and writes silu(g)*u (SwiGLU).
//
// Geometry: a wave-step covers 8 units of a row (8 lanes x 16 B each, contiguous in HBM); a wave owns MMVQ_R rows at
// a time and walks their k-steps one by one; the loads of step s+1 are in flight while step s is computed
// (two fragment sets), and the very first loads are issued before the activation prologue.
// weight rows per wave per pass: 2 (RoPE pairs live in one wave); 1 in DUAL mode, where a wave already carries two matrices
template <bool DUAL> struct rows_per_wave { static constexpr int R = DUAL ? 1 : 2; };

template <int TYPE, int T, int NW, bool DUAL, bool PRE>
__global__ void __launch_bounds__(NW*WAVE) k_mmvq(const mmvq_launch L) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool KQ = act_kind<TYPE>::K;
    constexpr int MMVQ_R = rows_per_wave<DUAL>::R;
    constexpr int NF = MMVQ_R * (DUAL ? 2 : 1);                 // fragments per k-step
    const int k = L.k;
    int8_t * lq = (int8_t *) smem;
    float  * ld = (float *)(smem + T*k);
    short  * lb = (short *)((char *) ld + T*(KQ ? k/256 : k/32)*4);
    char   * tail = (char *) lb + (KQ ? T*(k/16)*2 : 0);
    tail = (char *)(((uintptr_t) tail + 7) & ~(uintptr_t) 7);
    double * red = (double *) tail;
    float  * sc  = (float *)(tail + NW*T*8);

    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int nunits = wfrag<TYPE>::nunits(k);
    const int nsteps = (nunits + 7) / 8;
    int sub, alo, ahi, blo, bhi;
    wfrag<TYPE>::act_off(lane, sub, alo, ahi, blo, bhi);
    // group -> (matrix, first row)
    const int c0 = (L.m[0].rows + MMVQ_R - 1) / MMVQ_R;
    const int c1 = (!DUAL && L.n_mat > 1) ? (L.m[1].rows + MMVQ_R - 1) / MMVQ_R : 0;
    const int c2 = (!DUAL && L.n_mat > 2) ? (L.m[2].rows + MMVQ_R - 1) / MMVQ_R : 0;
    const int total = c0 + c1 + c2;
    const int gstride = gridDim.x*NW;

    const char * rp[NF];
    auto set_rows = [&](int g, int & mi, int & row0) {
        if (g < c0) { mi = 0; row0 = g*MMVQ_R; } else if (g < c0 + c1) { mi = 1; row0 = (g - c0)*MMVQ_R; } else { mi = 2; row0 = (g - c0 - c1)*MMVQ_R; }
#pragma unroll
        for (int r = 0; r < MMVQ_R; ++r) {
            rp[r] = L.m[mi].W + (size_t) min(row0 + r, L.m[mi].rows - 1) * L.m[mi].row_bytes;
            if (DUAL) rp[(DUAL ? MMVQ_R : 0) + r] = L.m[1].W + (size_t) min(row0 + r, L.m[1].rows - 1) * L.m[1].row_bytes;
        }
    };
    auto load_step = [&](wfrag<TYPE> * f, int s) {
        const int unit = min(s*8 + sub, nunits - 1);
#pragma unroll
        for (int i = 0; i < NF; ++i) f[i].load(rp[i], unit, lane, k);
    };

    // Two register stages of S k-steps each: while stage A is being consumed the loads of stage B are in flight
    // (2*S*NF 16-byte loads per lane outstanding: the bytes-in-flight that an HBM-bound stream needs at 8-16 waves/CU).
    constexpr int S = (TYPE == GGML_TYPE_Q6_K || TYPE == GGML_TYPE_Q5_K) ? 1 : 2;      // wider fragments: keep the register budget for occupancy
    wfrag<TYPE> fa[S][NF], fb[S][NF];
    auto load_stage = [&](wfrag<TYPE> (*f)[NF], int s0) {
#pragma unroll
        for (int u = 0; u < S; ++u) if (s0 + u < nsteps) load_step(f[u], s0 + u);
    };
    auto consume = [&](wfrag<TYPE> (*f)[NF], int s0, float (*acc)[T]) {
#pragma unroll
        for (int u = 0; u < S; ++u) {
            if (s0 + u >= nsteps) break;
            const int unit = (s0 + u)*8 + sub; const bool valid = wfrag<TYPE>::valid(unit, lane, k); const int uc = min(unit, nunits - 1);
#pragma unroll
            for (int i = 0; i < NF; ++i) f[u][i].decode(lane, valid);
#pragma unroll
            for (int t = 0; t < T; ++t) {
                act_regs a;
                const int8_t * aq = lq + t*k + uc*(KQ ? 256 : (TYPE == GGML_TYPE_Q8_0 ? 128 : 256));
                a.lo = *(const i32x4 *)(aq + alo); a.hi = *(const i32x4 *)(aq + ahi);
                if (KQ) { const short * bp = lb + t*(k/16) + uc*16; a.b0 = bp[blo]; a.b1 = bp[bhi]; a.dy = ld[t*(k/256) + uc]; }
                else    { a.b0 = a.b1 = 0; a.dy = ld[t*(k/32) + uc*(TYPE == GGML_TYPE_Q8_0 ? 4 : 8) + blo]; }
#pragma unroll
                for (int i = 0; i < NF; ++i) acc[i][t] += f[u][i].mac(a);
            }
        }
    };
    int g = blockIdx.x*NW + wave, mi = 0, row0 = 0;
    if (g < total) { set_rows(g, mi, row0); load_stage(fa, 0); }         // in flight across the prologue
    if (PRE) {                      // activations were quantised once by k_quant_act: copy the image (same layout) into LDS
        const int n16 = (int)((act_img_bytes(KQ, T, k) + 15) / 16);
        const i32x4 * src = (const i32x4 *) L.act.pre; i32x4 * dst = (i32x4 *) smem;
        for (int i = threadIdx.x; i < n16; i += NW*WAVE) dst[i] = src[i];
    } else {
        if (L.act.norm) row_scales<T, NW>(L.act, k, sc, red);
        if (KQ) quant_q8K_to_lds<T, NW>(L.act, sc, k, lq, ld, lb, wave, NW);
        else    quant_q80_to_lds<T, NW>(L.act, sc, k, lq, ld, wave, NW);
    }
    __syncthreads();

    while (g < total) {
        const mmvq_mat & M = L.m[mi];
        float acc[NF][T];
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][t] = 0.f;
        const int gn = g + gstride;
        for (int s = 0; s < nsteps; s += 2*S) {
            if (s + S < nsteps) load_stage(fb, s + S);
            consume(fa, s, acc);
            if (s + S >= nsteps) break;
            if (s + 2*S < nsteps) load_stage(fa, s + 2*S);
            consume(fb, s + S, acc);
        }
        // ---- next group's first loads go out before this group's reduction/epilogue
        const int cur_row0 = row0;
        int nmi = mi, nrow0 = row0;
        if (gn < total) { set_rows(gn, nmi, nrow0); load_stage(fa, 0); }
        // ---- reduce: every lane ends up with every sum
#pragma unroll
        for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][t] = wave_sum_f(acc[i][t]);
        // ---- epilogue: lane t finishes token t
        const bool ok1 = cur_row0 + 1 < M.rows;
        if (DUAL) {
#pragma unroll
            for (int t = 0; t < T; ++t) if (lane == t) {
                const float g0 = acc[0][t];
                *(float *)(L.m[0].out + (size_t) cur_row0*L.m[0].o_row + (size_t) t*L.m[0].o_tok) = (g0 / (1.0f + expf(-g0))) * acc[NF - 1][t];
            }
        } else if (M.epi == EPI_F32) {
#pragma unroll
            for (int t = 0; t < T; ++t) if (lane == t) {
                float v0 = acc[0][t], v1 = acc[NF > 1 ? 1 : 0][t];
                if (M.res) { v0 += M.res[(size_t) t*M.r_tok + cur_row0]; if (ok1) v1 += M.res[(size_t) t*M.r_tok + cur_row0 + 1]; }
                if (M.relu) { v0 = v0 > 0.f ? v0 : 0.f; v1 = v1 > 0.f ? v1 : 0.f; }
                *(float *)(M.out + (size_t) cur_row0*M.o_row + (size_t) t*M.o_tok) = v0;
                if (ok1) *(float *)(M.out + (size_t)(cur_row0 + 1)*M.o_row + (size_t) t*M.o_tok) = v1;
            }
        } else if (M.epi == EPI_F16) {
#pragma unroll
            for (int t = 0; t < T; ++t) if (lane == t) {
                *(__half *)(M.out + (size_t) cur_row0*M.o_row + (size_t) t*M.o_tok) = __float2half_rn(acc[0][t]);
                if (ok1) *(__half *)(M.out + (size_t)(cur_row0 + 1)*M.o_row + (size_t) t*M.o_tok) = __float2half_rn(acc[NF > 1 ? 1 : 0][t]);
            }
        } else {   // RoPE on the pair (row0, row0+1): theta by the reference's float recurrence (ggml_rope_cache_init)
            const int ip = (cur_row0 % L.rope.head_dim) >> 1;
#pragma unroll
            for (int t = 0; t < T; ++t) if (lane == t) {
                float theta = (float) L.rope.pos[t];
                for (int j = 0; j < ip; ++j) theta *= L.rope.theta_scale;
                const float th = L.rope.freq_scale * theta;
                const float c = cosf(th) * L.rope.attn_factor, sn = sinf(th) * L.rope.attn_factor;
                const float x0 = acc[0][t], x1 = acc[NF > 1 ? 1 : 0][t];
                const float y0 = x0*c - x1*sn, y1 = x0*sn + x1*c;
                if (M.epi == EPI_ROPE_F32) {
                    *(float *)(M.out + (size_t) cur_row0*M.o_row + (size_t) t*M.o_tok) = y0;
                    *(float *)(M.out + (size_t)(cur_row0 + 1)*M.o_row + (size_t) t*M.o_tok) = y1;
                } else {
                    *(__half *)(M.out + (size_t) cur_row0*M.o_row + (size_t) t*M.o_tok) = __float2half_rn(y0);
                    *(__half *)(M.out + (size_t)(cur_row0 + 1)*M.o_row + (size_t) t*M.o_tok) = __float2half_rn(y1);
                }
            }
        }
        g = gn; mi = nmi; row0 = nrow0;
    }
}

// Quantise-once: the same prologue as a stand-alone kernel writing the LDS image to HBM scratch; used when the
// image is big (T*k) so that 512 blocks do not each redo it.  Blocks share the units round-robin.
// The image holds Ttot tokens; this launch quantises tokens t0 .. t0+T-1 of it (a.X already points at token t0).
template <bool KQ, int T, int NW>
__global__ void __launch_bounds__(NW*WAVE) k_quant_act(const act_src a, int k, char * out, int Ttot, int t0) {
    __shared__ double red[NW*T];
    __shared__ float  sc[T];
    if (a.norm) row_scales<T, NW>(a, k, sc, red);
    const int nblk = KQ ? k/256 : k/32;
    int8_t * q = (int8_t *) out + (size_t) t0*k;
    float  * d = (float *)(out + (size_t) Ttot*k) + (size_t) t0*nblk;
    short  * bs = (short *)(out + (size_t) Ttot*k + (size_t) Ttot*nblk*4) + (size_t) t0*(k/16);
    const int wave = threadIdx.x / WAVE;
    if (KQ) {
        char * rec = out + act_img_bytes(true, Ttot, k);
        quant_q8K_to_lds<T, NW>(a, sc, k, q, d, bs, blockIdx.x*NW + wave, gridDim.x*NW, rec + (size_t) t0*nblk*32, rec + (size_t) Ttot*nblk*32 + (size_t) t0*nblk*32);
    } else quant_q80_to_lds<T, NW>(a, sc, k, q, d, blockIdx.x*NW + wave, gridDim.x*NW);
}
// ggml_silu_f32 (x / (1 + expf(-x))), the form kernels_ops.hip's silu_f and the SwiGLU epilogue of the tiled kernel use
__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }
// K-quant images, one wave per (token, super-block): with a folded RMS norm the wave sums the squares of ITS token row itself
// (16 KB at k = 4096, all loads in flight at once) and keeps the four values of its own super-block from that pass, so there is a
// single memory phase, no block barrier, and T*k/256 waves run in parallel (k_quant_act: every block first reduces all T rows).
__global__ void __launch_bounds__(512) k_quant_q8K(const act_src a, int k, char * out, int Ttot, int t0, int T) {
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int nsb = k/256, nu = T*nsb;
    int8_t * q = (int8_t *) out + (size_t) t0*k;
    float  * d = (float *)(out + (size_t) Ttot*k) + (size_t) t0*nsb;
    short  * bs = (short *)(out + (size_t) Ttot*k + (size_t) Ttot*nsb*4) + (size_t) t0*(k/16);
    char * rec = out + act_img_bytes(true, Ttot, k);
    char * rec32 = rec + (size_t) t0*nsb*32, * rec16 = rec + (size_t) Ttot*nsb*32 + (size_t) t0*nsb*32;
    for (int u = blockIdx.x*8 + wave; u < nu; u += gridDim.x*8) {
        const int t = u / nsb, sb = u - t*nsb;
        const float * row = a.X + (size_t) t*a.xs;
        float4 v;
        if (a.norm) {
            double s = 0.0; v = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int it0 = 0; it0*256 < k; it0 += 8) {               // 8 independent 16-byte loads per lane before the first use
                float4 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = (it0 + j)*256 + lane*4; x[j] = i < k ? *(const float4 *)(row + i) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s += (double)(x[j].x*x[j].x); s += (double)(x[j].y*x[j].y); s += (double)(x[j].z*x[j].z); s += (double)(x[j].w*x[j].w);
                    if (it0 + j == sb) v = x[j];
                }
            }
            const double tot = wave_sum_d(s);
            const float mean = (float)(tot / (double) k), sc = 1.0f / sqrtf(mean + a.eps);
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            if (a.norm_w) { const float4 w = *(const float4 *)(a.norm_w + sb*256 + lane*4); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
            if (a.norm_out) *(float4 *)(a.norm_out + (size_t) t*a.norm_os + sb*256 + lane*4) = v;
        } else {
            const int e = sb*256 + lane*4;
            v = (a.X2 && e >= a.ksplit) ? *(const float4 *)(a.X2 + (size_t) t*a.xs2 + (e - a.ksplit)) : *(const float4 *)(row + e);
            if (a.G) { const float4 g = *(const float4 *)(a.G + (size_t) t*a.gs + e); v.x *= act_silu(g.x); v.y *= act_silu(g.y); v.z *= act_silu(g.z); v.w *= act_silu(g.w); }     // SwiGLU folded in: silu(gate) * up
        }
        quant_q8K_unit(v, lane, t, sb, k, nsb, q, d, bs, rec32, rec16);
    }
}
// Q8_0 images of any number of tokens in one launch, one wave per (token, 256-element chunk = 8 blocks of 32), same single memory
// phase as k_quant_q8K.  (Round 2 filled such images 8 tokens per launch: nine launches per image at a 69-token verify batch, 30 % of
// the GPU time of a Q8_0 tree round - profiles/r03_verify_kernel_stats.txt.)
__global__ void __launch_bounds__(512) k_quant_q80(const act_src a, int k, char * out, int T) {
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int nch = (k + 255) / 256, nb = k / 32, nu = T*nch;
    int8_t * q = (int8_t *) out;
    float  * d = (float *)(out + (size_t) T*k);
    for (int u = blockIdx.x*8 + wave; u < nu; u += gridDim.x*8) {
        const int t = u / nch, ch = u - t*nch, e = ch*256 + lane*4;
        const float * row = a.X + (size_t) t*a.xs;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (a.norm) {
            double s = 0.0;
            for (int it0 = 0; it0*256 < k; it0 += 8) {
                float4 x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { const int i = (it0 + j)*256 + lane*4; x[j] = i < k ? *(const float4 *)(row + i) : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s += (double)(x[j].x*x[j].x); s += (double)(x[j].y*x[j].y); s += (double)(x[j].z*x[j].z); s += (double)(x[j].w*x[j].w);
                    if (it0 + j == ch) v = x[j];
                }
            }
            const double tot = wave_sum_d(s);
            const float mean = (float)(tot / (double) k), sc = 1.0f / sqrtf(mean + a.eps);
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            if (e < k) {
                if (a.norm_w) { const float4 w = *(const float4 *)(a.norm_w + e); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
                if (a.norm_out) *(float4 *)(a.norm_out + (size_t) t*a.norm_os + e) = v;
            }
        } else if (e < k) {
            v = (a.X2 && e >= a.ksplit) ? *(const float4 *)(a.X2 + (size_t) t*a.xs2 + (e - a.ksplit)) : *(const float4 *)(row + e);
            if (a.G) { const float4 g = *(const float4 *)(a.G + (size_t) t*a.gs + e); v.x *= act_silu(g.x); v.y *= act_silu(g.y); v.z *= act_silu(g.z); v.w *= act_silu(g.w); }
        }
        q80_unit(v, q + (size_t) t*k + e, d + (size_t) t*nb + e/32, lane, e < k);
    }
}
template <bool KQ> static void quant_act_T(hipStream_t st, int T, const act_src & a, int k, char * out, int Ttot, int t0) {
    if (KQ) {
        static const bool legacy = mi_lab_env("GGML_MI355X_QUANT_LEGACY") != nullptr;
        if (!legacy) { const int nu = T * (k/256); int grid = (nu + 7) / 8; if (grid > 512) grid = 512; k_quant_q8K<<<grid, 512, 0, st>>>(a, k, out, Ttot, t0, T); return; }
    }
    const int units = T * ((k + 255) / 256);
    int grid = (units + 8*2 - 1) / (8*2); if (grid > 64) grid = 64; if (grid < 1) grid = 1;
    switch (T) {
#define QA(n) case n: k_quant_act<KQ, n, 8><<<grid, 512, 0, st>>>(a, k, out, Ttot, t0); break;
        QA(1) QA(2) QA(3) QA(4) QA(5) QA(6) QA(7) QA(8)
#undef QA
        default: MI_ABORT("quant_act: T=%d", T);
    }
}
size_t mi_act_image_bytes(int type, int T, int k) { return (act_img_bytes_full(mi_traits(type).blck == 256, T, k) + 255) & ~(size_t) 255; }
void mi_quant_act(hipStream_t st, int type, int T, const act_src & a0, int k, char * out) {
    static const bool legacy = mi_lab_env("GGML_MI355X_QUANT_LEGACY") != nullptr;
    if (mi_traits(type).blck == 256 && !legacy) {             // one wave per (token, super-block): any number of tokens in one launch
        const int nu = T * (k/256); int grid = (nu + 7) / 8; if (grid > 1024) grid = 1024;
        k_quant_q8K<<<grid, 512, 0, st>>>(a0, k, out, T, 0, T);
        return;
    }
    if (mi_traits(type).blck == 32 && !legacy && k % 4 == 0) {
        const int nu = T * ((k + 255)/256); int grid = (nu + 7) / 8; if (grid > 1024) grid = 1024;
        k_quant_q80<<<grid, 512, 0, st>>>(a0, k, out, T);
        return;
    }
    for (int t0 = 0; t0 < T; t0 += 8) {                       // GGML_MI355X_QUANT_LEGACY: 8 tokens per launch (round 1)
        act_src a = a0; a.X += (size_t) t0 * a.xs;
        if (a.X2) a.X2 += (size_t) t0 * a.xs2;                 // the second CONCAT source advances with the tokens too
        if (a.norm_out) a.norm_out += (size_t) t0 * a.norm_os;
        const int n = T - t0 < 8 ? T - t0 : 8;
        if (mi_traits(type).blck == 256) quant_act_T<true>(st, n, a, k, out, T, t0); else quant_act_T<false>(st, n, a, k, out, T, t0);
    }
}

// ---------------------------------------------------------------- host side
static std::mutex g_attr_mu;
static std::unordered_map<const void *, std::pair<size_t, int>> g_occ;     // kernel -> (largest LDS asked for, resident blocks per CU at that size)
static int occupancy_blocks(const void * fn, int threads, size_t lds) {
    std::lock_guard<std::mutex> lk(g_attr_mu);
    auto it = g_occ.find(fn);
    if (it == g_occ.end() || it->second.first < lds) {
        int nb = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds));
        if (nb < 1) nb = 1;
        g_occ[fn] = { lds, nb };
        return nb;
    }
    return it->second.second;
}
static void ensure_lds_attr(const void * fn, size_t bytes) { if (bytes > 48*1024) mi_allow_big_lds(fn); }
// ---- optional HIP-event profile of every mat-vec launch (bench.py roofline; off by default, zero cost when off)
struct prof_rec { hipEvent_t a, b; double bytes; int T; double ops; };          // ops: int8 multiply-adds x 2 of the launch (2 T rows k)
static std::mutex g_prof_mu;
static bool g_prof_on = false;
// count-only mode (no events between the kernels: the launch sequence is the product's): launches and algorithmic bytes between
// count_begin / count_end, and a named marker kernel so that a rocprofv3 kernel trace of the same process can be cut at the same places
static bool g_count_on = false; static double g_count_bytes = 0; static long g_count_n = 0;
__global__ void k_profile_mark(int * p) { if (p && threadIdx.x == 0 && blockIdx.x == 0x7fffffff) *p = 0; }
extern "C" __attribute__((visibility("default"))) void ggml_backend_mi355x_count_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_count_on = true; g_count_bytes = 0; g_count_n = 0;
    k_profile_mark<<<1, 64, 0, hipStreamPerThread>>>(nullptr); (void) hipStreamSynchronize(hipStreamPerThread);
}
extern "C" __attribute__((visibility("default"))) long ggml_backend_mi355x_count_end(double * bytes) {
    (void) hipDeviceSynchronize();
    k_profile_mark<<<1, 64, 0, hipStreamPerThread>>>(nullptr); (void) hipStreamSynchronize(hipStreamPerThread);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_count_on = false; if (bytes) *bytes = g_count_bytes;
    return g_count_n;
}
static std::vector<prof_rec> g_prof, g_prof_cal;     // g_prof_cal: empty event pairs, the cost of the bracket itself
extern "C" __attribute__((visibility("default"))) void ggml_backend_mi355x_profile_begin(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto & r : g_prof) { (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    for (auto & r : g_prof_cal) { (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    g_prof.clear(); g_prof_cal.clear(); g_prof_on = true;
}
// out[0] = total kernel milliseconds, out[1] = total algorithmic bytes; returns the number of launches
extern "C" __attribute__((visibility("default"))) int ggml_backend_mi355x_profile_end(double * out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    double ms = 0, bytes = 0;
    for (auto & r : g_prof) { float t = 0; if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms += t; bytes += r.bytes; } (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    const int n = (int) g_prof.size();
    g_prof.clear();
    // out[2]: milliseconds an EMPTY event pair measures on the same stream (median of the calibration pairs): what the
    // bracket adds to every launch; bench.py reports kernel time net of it
    std::vector<float> cal;
    for (auto & r : g_prof_cal) { float t = 0; if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) cal.push_back(t); (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    g_prof_cal.clear();
    std::sort(cal.begin(), cal.end());
    if (out) { out[0] = ms; out[1] = bytes; out[2] = cal.empty() ? 0.0 : cal[cal.size()/2]; out[3] = (double) cal.size(); }
    return n;
}
// same as profile_end, the launches split by batch size: out[0..3] = {ms, bytes, int8 ops, launches} of the launches with fewer than
// min_tokens tokens, out[4..7] of the others (the big-batch GEMM when min_tokens = 25), out[8] = the empty event pair's ms
extern "C" __attribute__((visibility("default"))) int ggml_backend_mi355x_profile_end_by_batch(int min_tokens, double * out) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = false;
    double acc[8] = {};
    for (auto & r : g_prof) {
        float t = 0; const int o = r.T >= min_tokens ? 4 : 0;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { acc[o] += t; acc[o + 1] += r.bytes; acc[o + 2] += r.ops; acc[o + 3] += 1; }
        (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b);
    }
    const int n = (int) g_prof.size();
    g_prof.clear();
    std::vector<float> cal;
    for (auto & r : g_prof_cal) { float t = 0; if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) cal.push_back(t); (void) hipEventDestroy(r.a); (void) hipEventDestroy(r.b); }
    g_prof_cal.clear();
    std::sort(cal.begin(), cal.end());
    if (out) { for (int i = 0; i < 8; ++i) out[i] = acc[i]; out[8] = cal.empty() ? 0.0 : cal[cal.size()/2]; }
    return n;
}
// algorithmic bytes of one launch: weights once + fp32 activations once + outputs once
double mi_launch_bytes(const mmvq_launch & L, int T, bool dual) {
    double b = (double) T * L.k * 4;
    const int nm = dual ? 2 : L.n_mat;
    for (int i = 0; i < nm; ++i) b += (double) L.m[i].rows * L.m[i].row_bytes;
    if (dual) b += (double) L.m[0].rows * T * 4;
    else for (int i = 0; i < nm; ++i) b += (double) L.m[i].rows * T * ((L.m[i].epi == EPI_F16 || L.m[i].epi == EPI_ROPE_F16) ? 2 : 4);
    return b;
}

static inline size_t lds_total(bool ktype, int T, int k, int NW) { return act_lds_bytes(ktype, T, k) + 8 + (size_t) NW*T*8 + (size_t) T*4 + 16; }

// profile hooks shared with kernels_mmq.hip: returns a record index (or -1 when profiling is off)
int mi_prof_begin(hipStream_t st, const mmvq_launch & L, int T, bool dual) {
    if (g_count_on) { std::lock_guard<std::mutex> lk(g_prof_mu); g_count_bytes += mi_launch_bytes(L, T, dual); g_count_n++; }
    if (!g_prof_on) return -1;
    {   // every 64th launch: one empty pair right before, same stream, same queue state
        std::lock_guard<std::mutex> lk(g_prof_mu);
        if (g_prof.size() % 64 == 0) { prof_rec c; HIP_CHECK(hipEventCreate(&c.a)); HIP_CHECK(hipEventCreate(&c.b)); c.bytes = 0; c.T = 0; c.ops = 0; HIP_CHECK(hipEventRecord(c.a, st)); HIP_CHECK(hipEventRecord(c.b, st)); g_prof_cal.push_back(c); }
    }
    prof_rec r; HIP_CHECK(hipEventCreate(&r.a)); HIP_CHECK(hipEventCreate(&r.b)); r.bytes = mi_launch_bytes(L, T, dual);
    r.T = T; r.ops = 0; { const int nm = dual ? 2 : L.n_mat; for (int i = 0; i < nm; ++i) r.ops += 2.0 * T * (double) L.m[i].rows * L.k; }
    HIP_CHECK(hipEventRecord(r.a, st));
    std::lock_guard<std::mutex> lk(g_prof_mu); g_prof.push_back(r);
    return (int) g_prof.size() - 1;
}
void mi_prof_add_bytes(double bytes) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_count_on) g_count_bytes += bytes;
    if (g_prof_on && !g_prof.empty()) g_prof.back().bytes += bytes;
}
void mi_prof_end(hipStream_t st, int idx) {
    if (idx < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (idx < (int) g_prof.size()) HIP_CHECK(hipEventRecord(g_prof[idx].b, st));
}

template <int TYPE, int T, int NW, bool DUAL, bool PRE>
static void launch_one(hipStream_t st, const mmvq_launch & L) {
    const size_t lds = lds_total(act_kind<TYPE>::K, T, L.k, NW);
    MI_ASSERT(lds <= 160*1024);
    constexpr int MMVQ_R = rows_per_wave<DUAL>::R;
    int total = 0;
    if (DUAL) total = (L.m[0].rows + MMVQ_R - 1) / MMVQ_R;
    else for (int i = 0; i < L.n_mat; ++i) total += (L.m[i].rows + MMVQ_R - 1) / MMVQ_R;
    // Few, fat, persistent blocks: every block pays the activation prologue once, so the grid is sized to the
    // chip (as many 512-thread blocks per CU as registers and LDS allow), not to the row count; waves stride over row groups.
    auto fn = k_mmvq<TYPE, T, NW, DUAL, PRE>;
    ensure_lds_attr((const void *) fn, lds);
    int blocks_cu = occupancy_blocks((const void *) fn, NW*WAVE, lds);
    if (blocks_cu > 2) blocks_cu = 2;
    int grid = (total + NW - 1) / NW;
    const int cap = 256 * blocks_cu;
    if (grid > cap) grid = cap;
    if (grid < 1) return;
    const int pi = mi_prof_begin(st, L, T, DUAL);
    fn<<<grid, NW*WAVE, lds, st>>>(L);
    mi_prof_end(st, pi);
}
template <int TYPE, int T> static void launch_T(hipStream_t st, const mmvq_launch & L) {
    if (L.act.pre) { if (L.swiglu) launch_one<TYPE, T, 8, true, true >(st, L); else launch_one<TYPE, T, 8, false, true >(st, L); }
    else           { if (L.swiglu) launch_one<TYPE, T, 8, true, false>(st, L); else launch_one<TYPE, T, 8, false, false>(st, L); }
}
template <int TYPE> static void launch_type(hipStream_t st, int T, const mmvq_launch & L) {
    switch (T) {
        case 1: launch_T<TYPE, 1>(st, L); break; case 2: launch_T<TYPE, 2>(st, L); break;
        case 3: launch_T<TYPE, 3>(st, L); break; case 4: launch_T<TYPE, 4>(st, L); break;
        case 5: launch_T<TYPE, 5>(st, L); break; case 6: launch_T<TYPE, 6>(st, L); break;
        case 7: launch_T<TYPE, 7>(st, L); break; case 8: launch_T<TYPE, 8>(st, L); break;
        default: MI_ABORT("mmvq: T=%d", T);
    }
}

bool mi_mul_mat_q_supported_type(int type) {
    return type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K || type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q4_0;
}

int mi_mmvq_max_tokens(int type, int k) {
    const size_t per_tok = act_lds_bytes(mi_traits(type).blck == 256, 1, k);
    int tmax = (int) ((150*1024) / per_tok);
    return tmax > 8 ? 8 : tmax;
}

// Runs one (possibly multi-matrix) product over Ttot tokens, at most mi_mmvq_max_tokens() per launch.
// Activation images big enough to be worth quantising once (instead of in every block's prologue) go through a small
// ring of HBM scratch slots; entries are keyed by the ggml tensor that owns the activations (unique per graph_compute
// epoch), so wq|wk and a differently-typed wv, or gate|up launched separately, share one image.
static const int PRE_MIN_ELEMS = 8192;      // T*k above this: quantise once
// K-quants with >= this many tokens go to the matrix-core kernel (kernels_mmq.hip); GGML_MI355X_MMQ_MIN_T overrides (0 = never)
static int mmq_min_tokens() {
    static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMQ_MIN_T"); const int n = e ? atoi(e) : 2; return n <= 0 ? 1 << 30 : n; }();
    return v;
}
void mi_mmvq_run(hipStream_t st, int type, int Ttot, const mmvq_launch & L0, mi_act_cache * cache, const void * key) {
    if (L0.tiled) { mi_mmt_run(st, type, Ttot, L0, cache, key); return; }
    int tmax = mi_mmvq_max_tokens(type, L0.k);
    MI_ASSERT(tmax >= 1);
    const int kq = mi_traits(type).blck == 256;
    // big batches: the matrix-core kernel takes up to 32 tokens per pass (as many as its LDS image and the scratch slot allow)
    if (cache && cache->pool && Ttot > tmax && Ttot >= mmq_min_tokens()) {
        int tm = mi_mmq_max_tokens(type, L0.k, L0.swiglu != 0);
        while (tm > 8 && mi_act_image_bytes(type, tm, L0.k) > cache->slot_bytes) tm -= 8;
        if (tm > tmax && mi_mmq_supported(type, tm, L0.k, L0.swiglu != 0)) tmax = tm;
    }
    // k too long for the LDS image of the matrix-core kernel at the wanted tokens per pass (ffn_down at > 8 tokens, k = 28672 at any T >= 5):
    // split k into chunks of whole super-blocks, one launch per chunk, every later chunk adding to the output through the residual input.
    // Plain outputs only (no RoPE / f16 stores / SwiGLU, no folded norm: the row norm needs the whole row).
    if (kq && cache && cache->pool && L0.n_mat == 1 && !L0.swiglu && L0.m[0].epi == EPI_F32 && !L0.m[0].relu && !L0.act.norm && !L0.act.X2 && Ttot >= mmq_min_tokens()) {
        const int Tw = Ttot >= 24 ? 24 : (Ttot >= 16 ? 16 : (Ttot > 8 ? 8 : Ttot));
        if (!mi_mmq_supported(type, Tw, L0.k, false)) {
            const int nsb = L0.k / 256;
            int nch = 2, kc = 0;
            for (; nch <= 8; ++nch) { kc = (nsb + nch - 1) / nch * 256; if (mi_mmq_supported(type, Tw, kc, false) && mi_act_image_bytes(type, Tw, kc) <= cache->slot_bytes) break; }
            if (nch <= 8) {
                const size_t blk = mi_traits(type).size;
                for (int t0 = 0; t0 < Ttot; t0 += Tw) {
                    const int T = (Ttot - t0) < Tw ? (Ttot - t0) : Tw;
                    for (int k0 = 0, c = 0; k0 < L0.k; k0 += kc, ++c) {
                        mmvq_launch L = L0;
                        L.k = (L0.k - k0) < kc ? (L0.k - k0) : kc;
                        L.act.X += (size_t) t0 * L.act.xs + k0; L.act.pre = nullptr;
                        L.m[0].W += (size_t)(k0 / 256) * blk;
                        L.m[0].out += (size_t) t0 * L.m[0].o_tok;
                        if (c == 0) { if (L.m[0].res) L.m[0].res += (size_t) t0 * L.m[0].r_tok; }
                        else { L.m[0].res = (const float *) L.m[0].out; L.m[0].r_tok = L.m[0].o_tok / 4; }
                        const void * ckey = key ? (const void *)((const char *) key + k0 + 1) : nullptr;       // one image per (tensor, k-chunk)
                        int hit = -1;
                        if (ckey) for (int i = 0; i < MI_ACT_SLOTS; ++i) { const auto & e = cache->e[i]; if (e.key == ckey && e.epoch == cache->epoch && e.t0 == t0 && e.T == T && e.kq == kq && e.k == L.k) { hit = i; break; } }
                        if (hit < 0) {
                            hit = cache->next; cache->next = (cache->next + 1) % MI_ACT_SLOTS;
                            mi_quant_act(st, type, T, L.act, L.k, cache->pool + (size_t) hit * cache->slot_bytes);
                            cache->e[hit] = { ckey, cache->epoch, t0, T, kq, L.k };
                        }
                        L.act.pre = cache->pool + (size_t) hit * cache->slot_bytes;
                        mi_mmq_launch(st, type, T, L);
                    }
                }
                return;
            }
        }
    }
    for (int t0 = 0; t0 < Ttot; t0 += tmax) {
        const int T = (Ttot - t0) < tmax ? (Ttot - t0) : tmax;
        mmvq_launch L = L0;
        L.act.X += (size_t) t0 * L.act.xs;
        if (L.act.X2) L.act.X2 += (size_t) t0 * L.act.xs2;    // CONCAT(X, X2): both halves are token-major
        L.act.pre = nullptr;
        for (int i = 0; i < L.n_mat; ++i) { L.m[i].out += (size_t) t0 * L.m[i].o_tok; if (L.m[i].res) L.m[i].res += (size_t) t0 * L.m[i].r_tok; }
        if (L.rope.pos) L.rope.pos += t0;
        const bool mmq = cache && cache->pool && T >= mmq_min_tokens() && mi_mmq_supported(type, T, L.k, L.swiglu != 0) && mi_act_image_bytes(type, T, L.k) <= cache->slot_bytes;
        const bool inl = mmq && mi_mmq_inline_quant(type, T, L);
        if (!inl && cache && cache->pool && (mmq || (int64_t) T * L.k > PRE_MIN_ELEMS) && mi_act_image_bytes(type, T, L.k) <= cache->slot_bytes) {
            int hit = -1;
            if (key) for (int i = 0; i < MI_ACT_SLOTS; ++i) { const auto & e = cache->e[i]; if (e.key == key && e.epoch == cache->epoch && e.t0 == t0 && e.T == T && e.kq == kq && e.k == L.k) { hit = i; break; } }
            if (hit < 0) {
                hit = cache->next; cache->next = (cache->next + 1) % MI_ACT_SLOTS;
                mi_quant_act(st, type, T, L.act, L.k, cache->pool + (size_t) hit * cache->slot_bytes);
                cache->e[hit] = { key, cache->epoch, t0, T, kq, L.k };
            }
            L.act.pre = cache->pool + (size_t) hit * cache->slot_bytes;
        }
        { static const bool dbg = mi_lab_env("GGML_MI355X_DEBUG_MMQ") != nullptr; if (dbg) fprintf(stderr, "[mi355x] mat-vec type %d T %d (of %d, tmax %d) k %d -> %s\n", type, T, Ttot, tmax, L.k, mmq ? "mmq" : "mmvq"); }
        if (mmq) { mi_mmq_launch(st, type, T, L); continue; }
        switch (type) {
            case GGML_TYPE_Q4_K: launch_type<GGML_TYPE_Q4_K>(st, T, L); break;
            case GGML_TYPE_Q5_K: launch_type<GGML_TYPE_Q5_K>(st, T, L); break;
            case GGML_TYPE_Q6_K: launch_type<GGML_TYPE_Q6_K>(st, T, L); break;
            case GGML_TYPE_Q8_0: launch_type<GGML_TYPE_Q8_0>(st, T, L); break;
            case GGML_TYPE_Q4_0: launch_type<GGML_TYPE_Q4_0>(st, T, L); break;
            default: MI_ABORT("mmvq: unsupported weight type %d", type);
        }
    }
}

// dst[rows, T, b2, b3] = W[k, rows, b2/r2, b3/r3] . X[k, T, b2, b3]   (+ residual)
void mi_op_mul_mat_q(hipStream_t st, const ggml_tensor * dst, const ggml_tensor * residual, const ggml_tensor * out, mi_act_cache * cache, const act_src * act, const void * key) {
    const ggml_tensor * w = dst->src[0], * x = dst->src[1];
    const int k = (int) w->ne[0], rows = (int) w->ne[1];
    const int64_t Ttot = x->ne[1];
    if (k == 0 || rows == 0 || Ttot == 0) return;
    MI_ASSERT(x->type == GGML_TYPE_F32 && x->nb[0] == 4 && out->nb[0] == 4);
    const int64_t r2 = x->ne[2] / w->ne[2], r3 = x->ne[3] / w->ne[3];
    for (int64_t i3 = 0; i3 < x->ne[3]; ++i3) for (int64_t i2 = 0; i2 < x->ne[2]; ++i2) {
        mmvq_launch L{};
        L.act.X = (const float *)((const char *) x->data + i2*x->nb[2] + i3*x->nb[3]); L.act.xs = x->nb[1]/4;
        L.k = k; L.n_mat = 1;
        L.m[0].W = (const char *) w->data + (i2/r2)*w->nb[2] + (i3/r3)*w->nb[3]; L.m[0].row_bytes = w->nb[1]; L.m[0].rows = rows;
        L.m[0].epi = EPI_F32; L.m[0].out = (char *) out->data + i2*out->nb[2] + i3*out->nb[3]; L.m[0].o_row = 4; L.m[0].o_tok = out->nb[1];
        if (residual) { L.m[0].res = (const float *)((const char *) residual->data + i2*residual->nb[2] + i3*residual->nb[3]); L.m[0].r_tok = residual->nb[1]/4; }
        L.tiled = mi_ensure_tiled((ggml_tensor *) w) ? 1 : 0;              // eligible weights are re-laid out at their first use (tile_layout.h)
        if (act) { MI_ASSERT(x->ne[2] == 1 && x->ne[3] == 1 && L.tiled); L.act = *act; }
        mi_mmvq_run(st, w->type, (int) Ttot, L, cache, key ? key : ((x->ne[2] == 1 && x->ne[3] == 1) ? (const void *) x : nullptr));
    }
}
