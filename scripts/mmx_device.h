// mmx_device.h -- the quantised mat-vec of 1..8 tokens on tiled weights (round 3): k_mmx.
//
// Same arithmetic as k_mmt (kernels_mmt.hip) -- the CPU backend's Q8_K / Q8_0 activation quantisation
// (R/ggml/src/ggml-quants.c:194-215,2479-2512), integer dot products on v_mfma_i32_16x16x64_i8 (mmq_device.h) and the same
// fixed-order split-K reduction -- with a prologue rebuilt around what the phase stamps of round 2 showed
// (profiles/r02_mmt_phase_stamps.txt): the prologue is bound by VALU issue and dependent latencies, not by bytes.
//   * the signed extrema of a super-block are tracked on the integer pipe (a non-negative float orders like its bit pattern; among
//     negative floats the unsigned order is the magnitude order): v_max3_i32 / v_max3_u32 without NaN canonicalisation, and the
//     cross-lane steps are single DPP-fused integer max instructions;
//   * block sums (the mins / -32 offset records) are no longer reduced across lanes with 24 DPP adds per super-block: the wave
//     reads its own freshly written int8 bytes back from LDS, 32 consecutive per lane, and sums them with v_dot4 -- one lane
//     per (token, 32-group), records written with two byte stores;
//   * token slots are predicated once for all loads and once for all stores;
//   * a wave's weight tiles are ONE stream across its block's row groups, copied HBM -> LDS by LDS-DMA into a ring of slots with up to
//     four tiles in flight per wave (mmx_body); round 2 requested a row group's tiles when the previous group's units were done;
//   * no block barrier between the quantiser and the products: a wave multiplies exactly the super-blocks it quantised itself.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "mmq_device.h"
#include "tile_layout.h"

#define MX_RLD 33                       // f32x4 slots per wave in a reduction buffer (32 lanes + 1: spreads the readers over the banks)
#define MX_NSTAMP 12
#define MX_QSCRATCH 17424                // the quantiser's LDS scratch: 1 KiB of partial sums + 16 KiB of norm weights (its own region: a wave may
                                        // still be quantising when another one parks its first partial tile)

typedef float mxf4 __attribute__((ext_vector_type(4)));

// ---- the activation operands of ONE unit held in registers: a wave that multiplies the same unit of every row group (k <= 4096: one unit per
// wave and group) reads them from the LDS image once instead of once per row group, and the per-group predication of those reads goes too.
// mq_run_pre repeats mq_proc::run on them operation for operation (same results bit for bit).
template <int TYPE> struct mq_aop { static constexpr bool HAVE = false; };
template <int TYPE> struct mq_aop45 {
    static constexpr bool HAVE = true;
    i32x4 a[2][2], am; float dy[4];
    __device__ __forceinline__ void load(const mq_act & A, int sb, int lane) {
        const int i = lane & 15, kq = lane >> 4, tok_a = i & 7, cls_a = i >> 3;
        const bool av = cls_a == (kq >> 1) && tok_a < A.T;
        const int8_t * arow = A.q + tok_a*A.ldq + sb*256 + 128*cls_a + 16*(kq & 1);
#pragma unroll
        for (int ga = 0; ga < 2; ++ga) { a[ga][0] = av ? *(const i32x4 *)(arow + 64*ga) : (i32x4)(0); a[ga][1] = av ? *(const i32x4 *)(arow + 64*ga + 32) : (i32x4)(0); }
        am = (kq == 0 && tok_a < A.T) ? *(const i32x4 *)(A.rec + (tok_a*A.nsb + sb)*32 + 16*cls_a) : (i32x4)(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int tok = 4*(kq & 1) + r; dy[r] = tok < A.T ? A.d[tok*A.nsb + sb] : 0.f; }
    }
};
template <> struct mq_aop<GGML_TYPE_Q4_K> : mq_aop45<GGML_TYPE_Q4_K> {};
template <> struct mq_aop<GGML_TYPE_Q5_K> : mq_aop45<GGML_TYPE_Q5_K> {};
template <> struct mq_aop<GGML_TYPE_Q6_K> {
    static constexpr bool HAVE = true;
    i32x4 a[2][2], am; float dy[4];          // a[nn][nib]: the operand of this lane's own pass (p == qb); the other pass multiplies zeros
    __device__ __forceinline__ void load(const mq_act & A, int sb, int lane) {
        const int i = lane & 15, kq = lane >> 4, qb = kq >> 1, lh = kq & 1, tok_a = i & 7, cls_a = i >> 3;
        const bool av = cls_a == lh && tok_a < A.T;
        const int8_t * arow = A.q + tok_a*A.ldq + sb*256 + 16*lh;
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int nib = 0; nib < 2; ++nib) a[nn][nib] = av ? *(const i32x4 *)(arow + 128*nn + 32*(qb + 2*nib)) : (i32x4)(0);
        am = (kq == 0 && tok_a < A.T) ? *(const i32x4 *)(A.rec + (tok_a*A.nsb + sb)*32 + 16*cls_a) : (i32x4)(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { const int tok = 4*(kq & 1) + r; dy[r] = tok < A.T ? A.d[tok*A.nsb + sb] : 0.f; }
    }
};
template <int TYPE, class F> __device__ __forceinline__ void mq_run_pre(const F & f, const mq_aop<TYPE> & P, int lane, float (&acc)[1][4]) {
    const int kq = lane >> 4, g = kq;
    if constexpr (TYPE == GGML_TYPE_Q4_K || TYPE == GGML_TYPE_Q5_K) {
        const uint32_t u0 = f.hdr.y, u1 = f.hdr.z, u2 = f.hdr.w;
        const uint32_t s_lo = u0 & 0x3f3f3f3fu, s_hi = (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u);
        const uint32_t m_lo = u1 & 0x3f3f3f3fu, m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u);
        const float dw = h2f((uint16_t)(f.hdr.x & 0xffff)), mw = h2f((uint16_t)((uint32_t) f.hdr.x >> 16));
        const uint32_t sw = (g >> 1) ? s_hi : s_lo;
        int isum[4] = { 0, 0, 0, 0 };
#pragma unroll
        for (int ga = 0; ga < 2; ++ga) {
            i32x4 blo = f.qs[ga] & 0x0F0F0F0F, bhi = (f.qs[ga] >> 4) & 0x0F0F0F0F;
            if constexpr (TYPE == GGML_TYPE_Q5_K) {
                const i32x4 hb = (kq >> 1) ? (f.qh >> 4) : f.qh;
                blo |= ((hb >> (2*ga)) & 0x01010101) << 4; bhi |= ((hb >> (2*ga + 1)) & 0x01010101) << 4;
            }
            const int s0 = byte_of(sw, 2*ga), s1 = byte_of(sw, 2*ga + 1);
            const i32x4 c0 = mfma_i8(P.a[ga][0], blo);
            const i32x4 c1 = mfma_i8(P.a[ga][1], bhi);
#pragma unroll
            for (int r = 0; r < 4; ++r) isum[r] += __mul24(s0, c0[r]) + __mul24(s1, c1[r]);
        }
        i32x4 bm = {0, 0, 0, 0};
        if (kq == 0) { bm.x = (int) m_lo; bm.y = (int) m_hi; }
        const float mscale = (g >> 1) ? 128.f : 1.f;
        const i32x4 cm = mfma_i8(P.am, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][r] += (dw*P.dy[r])*(float) isum[r] - ((mw*P.dy[r])*mscale)*(float) cm[r];
    } else {
        static_assert(TYPE == GGML_TYPE_Q6_K, "mq_run_pre: K-quants only");
        const int qb = kq >> 1, lh = kq & 1, cls = g >> 1;
        int isum[4] = { 0, 0, 0, 0 };
        const i32x4 qh_own = f.get_qh(), scv = f.get_sc();
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) {
            const int src = 4*((lane & 15) + 16*(2*nn + lh));
            i32x4 qhn;
            qhn.x = __builtin_amdgcn_ds_bpermute(src, qh_own.x); qhn.y = __builtin_amdgcn_ds_bpermute(src, qh_own.y);
            qhn.z = __builtin_amdgcn_ds_bpermute(src, qh_own.z); qhn.w = __builtin_amdgcn_ds_bpermute(src, qh_own.w);
            const i32x4 hq = qhn >> (2*qb);
            const i32x4 qln = f.get_ql(nn);
#pragma unroll
            for (int nib = 0; nib < 2; ++nib) {
                const i32x4 b = nib ? (((qln >> 4) & 0x0F0F0F0F) | (hq & 0x30303030)) : ((qln & 0x0F0F0F0F) | ((hq << 4) & 0x30303030));
                const uint32_t sw = (uint32_t) scv[2*nn + nib] >> (8*cls);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int sc = sbyte_of(sw, 2*p);
                    const i32x4 a = qb == p ? P.a[nn][nib] : (i32x4)(0);
                    const i32x4 c = mfma_i8(a, b);
#pragma unroll
                    for (int r = 0; r < 4; ++r) isum[r] += __mul24(sc, c[r]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const i32x4 bm = kq == 0 ? scv : (i32x4)(0);
        const float dw = h2f((uint16_t) f.dh);
        const int mscale = (g >> 1) ? 128*32 : 32;
        const i32x4 cm = mfma_i8(P.am, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][r] += (dw*P.dy[r])*(float)(isum[r] - mscale*cm[r]);
    }
}


// launch flags (low to high): bits 0-1 reduction buffers (1 | 2), bits 4-5 pfpos: where the prologue requests the first weight tiles (0 right
// behind the activation loads | 1 once this wave's activations have landed | 2 once the norm scale is known / the image share is written)
#define MX_F_NBUF(f)  ((f) & 3)
#define MX_F_PFPOS(f) (((f) >> 4) & 3)
#define MX_F_DEPTH(f) (((f) >> 12) & 7)        // bits 12-14: ring slots per wave (tiles in flight), 1..4
#define MX_F_BATCH(f) (((f) >> 16) & 7)        // bits 16-18: row groups reduced behind one barrier, 1..MX_GMAX
#define MX_F_EPOCH(f) (((unsigned)(f) >> 20) & 0xfffu)   // bits 20-31: launch epoch (mx_landing_rank)
#define MX_GMAX 4

template <bool STAMP> __device__ __forceinline__ void mx_stamp(unsigned long long * base, int idx, int lane) {
    if constexpr (STAMP) { if (lane == 0 && base) base[idx] = __builtin_amdgcn_s_memrealtime(); }
}

// ---- cross-lane integer steps over the LG lanes of a token group (result in every lane of the group)
template <int LG> __device__ __forceinline__ int mx_grp_max_i(int v) {
    v = max(v, dpp_i<DPP_XOR1>(v)); v = max(v, dpp_i<DPP_XOR2>(v)); v = max(v, dpp_i<DPP_HMIR>(v));
    if (LG >= 16) v = max(v, dpp_i<DPP_MIR>(v));
    if (LG >= 32) { uint32_t a, b; lane_pair<16>((uint32_t) v, a, b); v = max((int) a, (int) b); }
    if (LG >= 64) { uint32_t a, b; lane_pair<32>((uint32_t) v, a, b); v = max((int) a, (int) b); }
    return v;
}
template <int LG> __device__ __forceinline__ uint32_t mx_grp_max_u(uint32_t v) {
    v = max(v, (uint32_t) dpp_i<DPP_XOR1>((int) v)); v = max(v, (uint32_t) dpp_i<DPP_XOR2>((int) v)); v = max(v, (uint32_t) dpp_i<DPP_HMIR>((int) v));
    if (LG >= 16) v = max(v, (uint32_t) dpp_i<DPP_MIR>((int) v));
    if (LG >= 32) { uint32_t a, b; lane_pair<16>(v, a, b); v = max(a, b); }
    if (LG >= 64) { uint32_t a, b; lane_pair<32>(v, a, b); v = max(a, b); }
    return v;
}
template <int LG> __device__ __forceinline__ int mx_grp_min_i(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v)); v = min(v, dpp_i<DPP_XOR2>(v)); v = min(v, dpp_i<DPP_HMIR>(v));
    if (LG >= 16) v = min(v, dpp_i<DPP_MIR>(v));
    if (LG >= 32) v = min_xw<16>(v);
    if (LG >= 64) v = min_xw<32>(v);
    return v;
}
template <int LG> __device__ __forceinline__ double mx_grp_sum_d(double v) {
    int2 p = *(int2 *) &v;
#define MX_DSTEP(C) { int2 q; q.x = dpp_i<C>(p.x); q.y = dpp_i<C>(p.y); v += *(double *) &q; p = *(int2 *) &v; }
    MX_DSTEP(DPP_XOR1) MX_DSTEP(DPP_XOR2) MX_DSTEP(DPP_HMIR)
    if (LG >= 16) MX_DSTEP(DPP_MIR)
#undef MX_DSTEP
    if (LG >= 32) v = sum_xw<16>(v);
    if (LG >= 64) v = sum_xw<32>(v);
    return v;
}
__device__ __forceinline__ const float * mx_act_ptr(const act_src & a, int t, int e) {
    return (a.X2 && e >= a.ksplit) ? a.X2 + (size_t) t*a.xs2 + (e - a.ksplit) : a.X + (size_t) t*a.xs + e;
}
__device__ __forceinline__ int mx_pack_b0(const i32x4 b) {       // low bytes of four dwords -> one dword
    return (int)(__builtin_amdgcn_perm((uint32_t) b.y, (uint32_t) b.x, 0x0c0c0400u) | __builtin_amdgcn_perm((uint32_t) b.w, (uint32_t) b.z, 0x04000c0cu));
}

// Lanes carry TOKENS (as in k_mmt): LG = 64, 32, 16 or 8 lanes serve one token (1, 2, <= 4, <= 8 tokens), lane = LG t + p; wave w owns
// super-blocks w, w + 16, ...  K-quants: lane p holds the float4s f = p + LG i (coalesced runs per token); Q8_0 / Q4_0: lane p holds
// 256 / LG consecutive elements.
template <int TYPE, int LG> struct mx_q {
    static constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    static constexpr int NF = 64/LG;
    static __device__ __forceinline__ constexpr int eoff(int i) { return Q80 ? 4*i : 4*LG*i; }
    static __device__ __forceinline__ int poff(int p) { return Q80 ? (256/LG)*p : 4*p; }

    static __device__ __forceinline__ void load(const act_src & a, mxf4 (&x)[NF], const bool tv, const int t, const int p, const int sb) {
        if (tv) {        // idle token slots are masked off: the vector memory pipe is paid per active lane
            const float * base = mx_act_ptr(a, t, sb*256) + poff(p);
#pragma unroll
            for (int i = 0; i < NF; ++i) x[i] = *(const mxf4 *)(base + eoff(i));
        } else {
#pragma unroll
            for (int i = 0; i < NF; ++i) x[i] = (mxf4){ 0.f, 0.f, 0.f, 0.f };
        }
    }

    // one super-block of every token slot of the wave -> int8 image, scales, split block sums
    static __device__ __forceinline__ void quant(const mxf4 (&x)[NF], const bool tv, const int t, const int p, const int lane, const int T, const int sb,
                                                 const int ldq, const int nsb, int8_t * lq, float * ldy, char * lrec) {
        int8_t * dst = lq + (size_t) t*ldq + sb*256 + poff(p);
        if constexpr (Q80) {
            // LG/8 consecutive lanes share a 32-element block: d = amax/127, id = 1/d, q = roundf(x*id), d kept through fp16 (quantize_row_q8_0_ref)
            int am = 0;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const i32x4 b = __builtin_bit_cast(i32x4, x[i]) & 0x7fffffff;
                am = max(max(b.x, b.y), am); am = max(max(b.z, b.w), am);
            }
            if (LG >= 16) am = max(am, dpp_i<DPP_XOR1>(am));
            if (LG >= 32) am = max(am, dpp_i<DPP_XOR2>(am));
            if (LG >= 64) am = max(am, dpp_i<DPP_HMIR>(am));
            const float amax = __int_as_float(am);
            const float dd = amax / 127.f;
            const float id = dd ? 1.0f/dd : 0.0f;
            int pk[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int q0 = (int) roundf(x[i].x*id), q1 = (int) roundf(x[i].y*id), q2 = (int) roundf(x[i].z*id), q3 = (int) roundf(x[i].w*id);
                pk[i] = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | (q3 << 24);
            }
            if (tv) {
#pragma unroll
                for (int i = 0; i < NF; ++i) *(int *)(dst + 4*i) = pk[i];
                if ((p % (LG/8)) == 0) ldy[t*nsb + sb*8 + p / (LG/8)] = __half2float(__float2half_rn(dd));
            }
        } else {
            // Q8_K rule (quantize_row_q8_K_ref): the scale comes from the FIRST element of largest magnitude.  pmb = bits of max(+x) (signed
            // integer order, floor +0), nmb = bits of the most negative element (unsigned order, floor -0)
            int pmb = 0; uint32_t nmb = 0x80000000u;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const i32x4 b = __builtin_bit_cast(i32x4, x[i]);
                pmb = max(max(b.x, b.y), pmb); pmb = max(max(b.z, b.w), pmb);
                nmb = max(max((uint32_t) b.x, (uint32_t) b.y), nmb); nmb = max(max((uint32_t) b.z, (uint32_t) b.w), nmb);
            }
            pmb = mx_grp_max_i<LG>(pmb); nmb = mx_grp_max_u<LG>(nmb);
            const float pm = __int_as_float(pmb), nm = __uint_as_float(nmb & 0x7fffffffu);
            const float amax = pmb > (int)(nmb & 0x7fffffffu) ? pm : nm;
            float mx = pmb > (int)(nmb & 0x7fffffffu) ? pm : -nm;
            const bool tie = pmb == (int)(nmb & 0x7fffffffu) && pmb != 0;
            if (__any(tie)) {      // a positive and a negative element share the largest magnitude: the first one in element order decides
                int key = 0x7fffffff;
#pragma unroll
                for (int i = NF - 1; i >= 0; --i) {
                    const float xe[4] = { x[i].x, x[i].y, x[i].z, x[i].w };
#pragma unroll
                    for (int e = 3; e >= 0; --e) if (fabsf(xe[e]) == amax) key = ((4*(p + LG*i) + e) << 1) | (xe[e] < 0.f ? 1 : 0);
                }
                key = mx_grp_min_i<LG>(key);
                if (tie) mx = (key & 1) ? -amax : amax;
            }
            const bool nz = pmb != 0 || (nmb & 0x7fffffffu) != 0;
            const float iscale = nz ? -127.f / mx : 0.f;
            const float dd = nz ? 1.0f / iscale : 0.f;
            int pk[NF];
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                mxf4 r;
                {   // rounded product, THEN nearest_int's magic addition (ggml-quants.c:559-565): the two roundings must not be contracted into an FMA
#pragma clang fp contract(off)
                    const mxf4 m = x[i] * iscale;
                    r = m + 12582912.f;
                }
                pk[i] = mx_pack_b0(__builtin_bit_cast(i32x4, r));
            }
            if (tv) {
#pragma unroll
                for (int i = 0; i < NF; ++i) *(int *)(dst + 4*LG*i) = pk[i];
                if (p == 0) ldy[t*nsb + sb] = dd;
            }
            // block sums for the mins / offset MFMA, split as s = 128 h + l: the wave reads its own bytes back (LDS is in order per wave), lane
            // (token tt, 32-group g) sums 32 consecutive int8 with dot4; 8 * 64/LG lanes take part
            const int tt = lane >> 3, g = lane & 7;
            if (tt < 64/LG && tt < T) {
                const int8_t * src = lq + (size_t) tt*ldq + sb*256 + 32*g;
                const i32x4 a0 = *(const i32x4 *) src, a1 = *(const i32x4 *)(src + 16);
                const int s0 = dot16(a0, (i32x4)(0x01010101)), s1 = dot16(a1, (i32x4)(0x01010101));
                int8_t * rec = (int8_t *) lrec + (size_t)(tt*nsb + sb)*32;
                if constexpr (TYPE == GGML_TYPE_Q6_K) {     // rec16: [l0..l15, h0..h15], 16-element groups 2g, 2g + 1
                    *(uint16_t *)(rec + 2*g)      = (uint16_t)((s0 & 127) | ((s1 & 127) << 8));
                    *(uint16_t *)(rec + 16 + 2*g) = (uint16_t)(((s0 >> 7) & 0xff) | (((s1 >> 7) & 0xff) << 8));
                } else {                                    // rec32: [l0..l7, 0 x 8, h0..h7, 0 x 8]
                    const int s = s0 + s1;
                    rec[g] = (int8_t)(s & 127); rec[16 + g] = (int8_t)(s >> 7);
                    if (g < 4) *(int *)(rec + 8 + 4*(g & 1) + 16*(g >> 1)) = 0;
                }
            }
        }
    }
};

// Order in which the waves of a block pass a point, without a barrier and without initialised LDS: the word holds (launch epoch << 8 | count);
// the first wave of a launch finds another epoch and restarts the count.  The rank only steers WHEN a wave requests its tiles -- a stale word that
// happens to carry this launch's 12-bit epoch costs time, never correctness.
__device__ __forceinline__ int mx_landing_rank(unsigned * word, unsigned epoch, int lane) {
    int rank = 0;
    if (lane == 0) {
        unsigned old = *(volatile unsigned *) word;
        for (;;) {
            const bool mine = (old >> 8) == epoch;
            const unsigned want = mine ? old + 1 : ((epoch << 8) | 1u);
            const unsigned seen = atomicCAS(word, old, want);
            if (seen == old) { rank = mine ? (int)(old & 0xffu) : 0; break; }
            old = seen;
        }
    }
    return __builtin_amdgcn_readfirstlane(rank);
}
// `prefetch` issues the block's first weight-tile loads; pfpos says where (MX_F_PFPOS)
template <int TYPE, int LG, int NSB, bool STAMP, class PFN>
__device__ __forceinline__ void mx_norm_quant(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                              int8_t * lq, float * ldy, char * lrec, double * rd, const int lane, const int wave, PFN prefetch, const int pfpos,
                                              unsigned long long * stp) {
    using Q = mx_q<TYPE, LG>;
    constexpr int NF = Q::NF;
    const int t = lane / LG, p = lane % LG;
    const bool tv = t < T;
    mxf4 xv[NSB][NF], wraw;
#pragma unroll
    for (int c = 0; c < NSB; ++c) if (wave + 16*c < nun) Q::load(a, xv[c], tv, t, p, wave + 16*c);
    // norm weights of this wave's super-block (NSB == 1): ONE coalesced 1 KiB request, handed to the token groups through the wave's 1 KiB of LDS
    const bool wlds = NSB == 1 && a.norm_w && wave < nun;
    mxf4 * wst = (mxf4 *)((char *) rd + 1024) + wave*64;
    if (wlds) wraw = *(const mxf4 *)(a.norm_w + wave*256 + 4*lane);
    if constexpr (STAMP) { if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mx_stamp<STAMP>(stp, 7, lane); } }
    if (pfpos == 0) prefetch();
    if (pfpos == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); prefetch(); }       // this wave's activations are here: the weight requests no longer queue in front of them
    // ggml_compute_forward_rms_norm_f32 (R/ggml/src/ggml-cpu/ggml-cpu.c:7098-7144): sum of x*x (float products) in double
    double ss = 0.0;
#pragma unroll
    for (int c = 0; c < NSB; ++c) {
        if (wave + 16*c < nun) {
#pragma unroll
            for (int i = 0; i < NF; ++i) { const mxf4 q = xv[c][i] * xv[c][i]; ss += (double) q.x; ss += (double) q.y; ss += (double) q.z; ss += (double) q.w; }
        }
    }
    ss = mx_grp_sum_d<LG>(ss);
    if (p == 0 && tv) rd[t*16 + wave] = ss;
    if (wlds) wst[lane] = wraw;
    mx_stamp<STAMP>(stp, 8, lane);
    __syncthreads();
    mx_stamp<STAMP>(stp, 9, lane);
    double tot = 0.0;
    if (tv) {
        if (LG >= 16) { if (p < 16) tot = rd[t*16 + p]; }
        else tot = rd[t*16 + p] + rd[t*16 + p + 8];
    }
    tot = mx_grp_sum_d<LG>(tot);
    const float mean = (k & (k - 1)) == 0 ? (float) __builtin_ldexp(tot, -__builtin_ctz(k)) : (float)(tot / (double) k);
    const float s1 = 1.0f / sqrtf(mean + a.eps);
    mx_stamp<STAMP>(stp, 10, lane);
    if (pfpos >= 2) prefetch();              // (3 = adaptive in the plain path; behind the norm barrier every wave's activations have landed)
#pragma unroll
    for (int c = 0; c < NSB; ++c) {
        const int sb = wave + 16*c;
        if (sb < nun) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int fi = Q::Q80 ? NF*p + i : p + LG*i;                    // float4 of the super-block this register holds
                const int e = sb*256 + 4*fi;
                mxf4 v = xv[c][i] * s1;
                if (a.norm_w) v *= (NSB == 1) ? wst[fi] : *(const mxf4 *)(a.norm_w + e);
                if (a.norm_out && blockIdx.x == 0 && tv) *(mxf4 *)(a.norm_out + (size_t) t*a.norm_os + e) = v;
                xv[c][i] = v;
            }
            Q::quant(xv[c], tv, t, p, lane, T, sb, ldq, nsb, lq, ldy, lrec);
        }
    }
}

template <int TYPE, int LG, bool STAMP, class PFN>
__device__ __forceinline__ void mx_quantise_lg(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                               int8_t * lq, float * ldy, char * lrec, double * rd, const int lane, const int wave, PFN prefetch, const int pfpos,
                                               const unsigned epoch, unsigned long long * stp) {
    using Q = mx_q<TYPE, LG>;
    constexpr int NF = Q::NF;
    const int t = lane / LG, p = lane % LG;
    const bool tv = t < T;
    if (a.norm) {
        if (nun <= 16) mx_norm_quant<TYPE, LG, 1, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
        else           mx_norm_quant<TYPE, LG, 2, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
    } else {
        // (keeping a second super-block of activations in flight per wave costs 32 registers at 8 token slots: the tile ring needs them)
        mxf4 x[NF];
        if (wave < nun) Q::load(a, x, tv, t, p, wave);
        if constexpr (STAMP) { if (stp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mx_stamp<STAMP>(stp, 7, lane); } }
        if (pfpos == 0) prefetch();
        if (pfpos == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); prefetch(); }
        bool late = false;
        if (pfpos == 3) {
            // The block waits for its slowest wave at the first reduction, and that wave's chain is landing -> quantiser -> tile round trip ->
            // products.  A wave whose activations land among the last eight of the block requests its tiles at once (the others' activation
            // loads are done: nothing of theirs queues behind the requests any more); an early wave has the slack to ask after its quantiser
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            late = mx_landing_rank((unsigned *)((char *) rd + 17408), epoch, lane) >= 8;
            if (late) prefetch();
        }
        for (int sb = wave; sb < nun; sb += 16) {
            if (sb != wave) Q::load(a, x, tv, t, p, sb);
            Q::quant(x, tv, t, p, lane, T, sb, ldq, nsb, lq, ldy, lrec);
        }
        if (pfpos == 2 || (pfpos == 3 && !late)) prefetch();
    }
}
template <int TYPE, bool STAMP, class PFN>
__device__ __forceinline__ void mx_quantise(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                            int8_t * lq, float * ldy, char * lrec, double * rd, const int lane, const int wave, PFN prefetch, const int pfpos,
                                            const unsigned epoch, unsigned long long * stp) {
    if (T == 1)      mx_quantise_lg<TYPE, 64, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, epoch, stp);
    else if (T == 2) mx_quantise_lg<TYPE, 32, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, epoch, stp);
    else if (T <= 4) mx_quantise_lg<TYPE, 16, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, epoch, stp);
    else             mx_quantise_lg<TYPE, 8, STAMP>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, epoch, stp);
}

// {cos, sin} of one rotation without the per-forward table (graphs with fewer than four ROPE nodes): theta by the reference's float recurrence
// (ggml_rope_cache_init).  Not inlined: the epilogue is instantiated once per ring slot.
__device__ __noinline__ float2 mx_rope_cs(float th, int ip, float theta_scale, float freq_scale, float attn_factor) {
    for (int j = 0; j < ip; ++j) th *= theta_scale;
    const float a = freq_scale * th;
    return make_float2(cosf(a) * attn_factor, sinf(a) * attn_factor);
}
static inline size_t mx_img_bytes(int T, int k, bool q80) {
    const size_t nsc = q80 ? k/32 : k/256;
    return (size_t) T*(k + 16) + (((size_t) T*nsc*4 + 15) & ~(size_t) 15) + (q80 ? 0 : (size_t) T*nsc*32);
}
static inline size_t mx_lds_bytes(int T, int k, bool dual, bool q80, int nbuf, int batch = 1) {
    const size_t red = (size_t) nbuf * batch * (dual ? 2 : 1) * 16 * MX_RLD * 16;
    return mx_img_bytes(T, k, q80) + MX_QSCRATCH + red;                // image | quantiser scratch | reduction buffers (| tile ring: mx_ring_depth)
}
// ring slots per wave that fit behind the rest (at most `want`, at least 1; 0 = does not fit at all)
static inline int mx_ring_depth(int T, int k, bool dual, bool q80, int nbuf, int tile, int want, int batch = 1) {
    const size_t base = mx_lds_bytes(T, k, dual, q80, nbuf, batch);
    if (base + (size_t) 16 * tile > 160*1024) return 0;
    int d = (int)((160*1024 - base) / ((size_t) 16 * tile));
    return d < want ? d : want;
}

// ---- the weight stream through LDS-DMA.  A wave copies its next tiles HBM -> LDS with global_load_lds_dwordx4 (1 KiB per wave-instruction, no
// register destination), D of them in flight in its own ring of LDS slots, and waits for the oldest one with a counted s_waitcnt vmcnt.
// Nothing of a tile in flight lives in a register: a register ring made hipcc copy fragments at the loop's back edge (and wait for the
// newest request to land first).  The statements are inline asm: hipcc does not count them, its own waits stay correct (vmcnt retires in
// order; they can only over-wait), and the waits here are counted from the tiles requested behind the one that is needed.
// LDS destination = M0 + lane * 16 (wave-uniform base, lane-linear): a slot holds the tile byte for byte as HBM does.
template <bool NT = true> __device__ __forceinline__ void mx_dma16(const char * gsrc, uint32_t lds_dst) {      // all active lanes: 16 bytes from gsrc to lds_dst + 16 lane
    unsigned keep;
    if constexpr (NT) asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else              asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                                   : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// What the NEXT mat-vec launch of the graph will stream first (its row groups in tile order): a block whose own stream is exhausted keeps its
// ring busy with those bytes -- default cache policy, never read from LDS -- so that they wait in L2 / the Infinity Cache when the next
// launch asks for them 3 us into its prologue, instead of costing it an HBM round trip under a chip-wide burst.
struct mx_next { const char * w; int group_bytes; int groups; };
template <int TILE> __device__ __forceinline__ void mx_dma_tile(const char * tile, uint32_t slot, int lane) {
    constexpr int FULL = TILE / 1024, REST = (TILE % 1024) / 16;
    static_assert(TILE % 16 == 0, "tiles are whole 16-byte pieces");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                       // the slot's previous tenant has been read into registers
#pragma unroll
    for (int c = 0; c < FULL; ++c) mx_dma16(tile + c*1024 + lane*16, slot + c*1024);
    if (REST) { if (lane < REST) mx_dma16(tile + FULL*1024 + lane*16, slot + FULL*1024); }
}
template <int TILE> __device__ __forceinline__ constexpr int mx_dma_ops() { return TILE / 1024 + ((TILE % 1024) ? 1 : 0); }
// wait until at most n of this wave's vector-memory operations are outstanding (n is wave-uniform, 0..20)
__device__ __forceinline__ void mx_wait_vm(int n) {
    switch (n) {
#define MX_W(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
        MX_W(0) MX_W(1) MX_W(2) MX_W(3) MX_W(4) MX_W(5) MX_W(6) MX_W(7) MX_W(8) MX_W(9) MX_W(10) MX_W(11) MX_W(12) MX_W(13) MX_W(14) MX_W(15) MX_W(16) MX_W(17) MX_W(18) MX_W(19) MX_W(20)
#undef MX_W
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// L.m[i] for a run-time (wave-uniform) i: read through the kernel-argument segment with scalar loads.  Indexing the by-value argument itself
// with a run-time i makes the compiler copy all of it to scratch.  KOFF = byte offset of the mmvq_launch inside the kernel's arguments.
typedef const mmvq_mat __attribute__((address_space(4))) * mx_matp;
template <int KOFF> __device__ __forceinline__ mx_matp mx_mat(int i) {
    const char __attribute__((address_space(4))) * base = (const char __attribute__((address_space(4))) *) __builtin_amdgcn_kernarg_segment_ptr();
    return (mx_matp)(base + KOFF + offsetof(mmvq_launch, m)) + i;
}

// the kernel body: block `bid` of `nblk` (k_mmx: the whole grid; k_mmx2: one of the two partitions of a mixed-type launch).  16 waves.
// A block owns the row groups bid, bid + nblk, ...; its 16 waves split the units of a group (split-K).  What a wave reads is ONE stream of
// tiles -- (group 0, its units), (group 1, its units), ... -- D of them in flight in its LDS ring, across the group boundaries: round 2
// requested a group's tiles only when the previous group's units were done, i.e. one exposed HBM round trip per group (three per q|k|v
// launch at k = 4096, where a wave has a single unit per group).
// DUAL: gate | up with the SwiGLU epilogue; a group's stream elements alternate gate, up.
template <int TYPE, bool DUAL, bool HOIST, bool STAMP, int KOFF>
__device__ __forceinline__ void mmx_body(const mmvq_launch & L, const int T, const int flags, const int bid, const int nblk, unsigned long long * stamp_base, const mx_next nx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NW = 16;
    const int nbuf = MX_F_NBUF(flags), pfpos = MX_F_PFPOS(flags);
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int NM = DUAL ? 2 : 1;
    constexpr int TILE = mq_tfrag<TYPE>::TILE;
    const int k = L.k, nun = k/256, nsb = Q80 ? k/32 : nun, ldq = k + 16;
    int8_t * lq  = (int8_t *) smem;
    float  * ldy = (float *)(smem + (size_t) T*ldq);
    char   * lrec = (char *) ldy + (((size_t) T*nsb*4 + 15) & ~(size_t) 15);
    double * qscr = (double *)(lrec + (Q80 ? 0 : (size_t) T*nsb*32));
    f32x4  * red = (f32x4 *)((char *) qscr + MX_QSCRATCH);
    const int D = MX_F_DEPTH(flags);
    char   * ring = (char *) red + (size_t) nbuf * MX_F_BATCH(flags) * NM * NW * MX_RLD * 16;          // [wave][slot][TILE]
    // the wave index through readfirstlane: everything derived from it (stream cursors, row groups, matrix index) is provably wave-uniform and
    // lives in scalar registers -- with a divergent-looking index into L.m[] the compiler copies the whole kernel argument to scratch
    const int tid = threadIdx.x, lane = tid % WAVE, wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    unsigned long long * stp = nullptr;
    if constexpr (STAMP) { if (stamp_base && blockIdx.x < 256) stp = stamp_base + ((size_t) blockIdx.x * 16 + wave) * MX_NSTAMP; }
    mx_stamp<STAMP>(stp, 0, lane);

    const int c0 = L.m[0].rows >> 4;
    const int c1 = (!DUAL && L.n_mat > 1) ? L.m[1].rows >> 4 : 0;
    const int c2 = (!DUAL && L.n_mat > 2) ? L.m[2].rows >> 4 : 0;
    const int total = c0 + c1 + c2;
    const int nu = wave < nun ? ((nun - wave + NW - 1) / NW) * NM : 0;     // stream elements ((matrix,) unit) of this wave per row group

    auto rows_of = [&](int g, int & mi, int & row0) { if (g < c0) { mi = 0; row0 = g*16; } else if (g < c0 + c1) { mi = 1; row0 = (g - c0)*16; } else { mi = 2; row0 = (g - c0 - c1)*16; } };
    // ---- load cursor: the next stream element to request
    int lgrp = bid, lu = 0;
    const char * ltp[NM];
    // (L.m[] is indexed with constants only; a run-time index goes through the kernel-argument segment, mx_mat)
    const int64_t rowb = L.m[0].row_bytes;       // one weight type: one row size
    auto lrows = [&]() { int mi, row0; rows_of(lgrp, mi, row0); ltp[0] = mx_mat<KOFF>(mi)->W + (size_t) row0 * rowb; if (DUAL) ltp[NM - 1] = L.m[1].W + (size_t) row0 * rowb; };
    if (lgrp < total) lrows();
    int rslot = 0, issued = 0, consumed = 0;                                      // ring: slot of the next request; tiles requested / multiplied so far
    char * const myring = ring + (size_t) wave * D * TILE;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *) myring;     // LDS byte address of this wave's slot 0
    auto request_next = [&]() {      // DMA of the stream element at the load cursor into the next ring slot
        const int unit = DUAL ? wave + (lu >> 1)*NW : wave + lu*NW;
        mx_dma_tile<TILE>((DUAL ? ltp[lu & 1] : ltp[0]) + (size_t) unit * TILE, __builtin_amdgcn_readfirstlane(ring_lds + rslot * TILE), lane);
        rslot = rslot + 1 == D ? 0 : rslot + 1; ++issued;
        if (++lu == nu) { lu = 0; lgrp += nblk; if (lgrp < total) lrows(); }
    };
    // phantom requests (mx_next): 1 KiB chunks w, w + 16, ... of row group blockIdx.x of the next launch, as many as fit the slot just freed
    int ph_ops = 0, ph_chunk = wave;
    const int ph_n = (nx.w && (int) blockIdx.x < nx.groups) ? (nx.group_bytes + 1023) >> 10 : 0;
    const char * const ph_base = nx.w + (size_t) blockIdx.x * nx.group_bytes;
    auto phantom_into = [&](int slot) {
#pragma unroll
        for (int c = 0; c < TILE / 1024; ++c) {
            if (ph_chunk < ph_n) {
                const int left = nx.group_bytes - ph_chunk*1024;        // a group is whole 16-byte pieces
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane*16 < left) mx_dma16<false>(ph_base + (size_t) ph_chunk*1024 + lane*16, __builtin_amdgcn_readfirstlane(ring_lds + slot*TILE + c*1024));
                ph_chunk += NW; ++ph_ops;
            }
        }
    };
    auto prefetch = [&]() {          // the first D tiles of the stream are in flight across the prologue
        for (int j = 0; j < D; ++j) if (nu > 0 && lgrp < total) request_next();
    };

    mx_quantise<TYPE, STAMP>(L.act, T, k, nun, nsb, ldq, lq, ldy, lrec, qscr, lane, wave, prefetch, pfpos, MX_F_EPOCH(flags), stp);
    mx_stamp<STAMP>(stp, 1, lane);                                 // this wave's share of the activation image is in LDS
    // No block barrier here: wave w multiplies exactly the super-blocks w, w + 16, ... it has just quantised itself (the units of its split-K
    // share), for every row group -- the image is wave-private; LDS is in order per wave
    if (flags & 0x100) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    mx_stamp<STAMP>(stp, 2, lane);
    const mq_act A = { lq, ldq, ldy, lrec, nsb, T };

    int par = 0, cgi = 0; bool first = true;
    float acc[NM][1][4];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[m][0][r] = 0.f;
    // ---- split-K reduction + epilogues, G row groups behind ONE barrier.
    // park(j): a wave adds its two classes (lanes l, l ^ 32) and parks 32 x f32x4 -- partial sums of (row n = l & 15, tokens 4 (l >> 4) .. + 3) of
    // the batch's j-th group -- right behind that group's units, and goes on with the next group's tiles.  reduce_batch(): behind the barrier all
    // 16 waves reduce: a DPP row of 16 lanes holds the 16 waves' partials of (row n, token pair 4h + 2pr ..) of one group, four rows per wave,
    // and the row's first two lanes finish one token each -- round 2 synchronised after every group and left the sums to waves 0..7, which
    // then ran late into the next group.  What an epilogue needs from memory (residual, RoPE table) is requested before the barrier.
    const int G = MX_F_BATCH(flags);
    auto park = [&](int j) {
        f32x4 * rb = red + (size_t)(par*G + j) * NM * NW * MX_RLD;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            f32x4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = sum_xw<32>(acc[m][0][r]); acc[m][0][r] = 0.f; }
            if (lane < 32) rb[(m*NW + wave)*MX_RLD + lane] = v;
        }
    };
    auto reduce_batch = [&](int nb) {          // the batch's groups are bid + (cgi + j) nblk, j < nb
        if (first) mx_stamp<STAMP>(stp, 3, lane);
        const int qg = wave*4 + (lane >> 4), w = lane & 15;
        const int n = qg & 15, h = (qg >> 4) & 1, pr = qg >> 5;
        const int tok = 4*h + 2*pr + (w & 1);
        const bool act = w < 2 && tok < T;
        int row[MX_GMAX], cmi[MX_GMAX]; float resv[MX_GMAX]; float2 cs[MX_GMAX];
#pragma unroll
        for (int j = 0; j < MX_GMAX; ++j) {
            resv[j] = 0.f; cs[j] = make_float2(1.f, 0.f); row[j] = 0; cmi[j] = 0;
            if (j < nb) {
                int crow0; rows_of(bid + (cgi + j)*nblk, cmi[j], crow0); row[j] = crow0 + n;
                if (!DUAL && act) {
                    const mx_matp M = mx_mat<KOFF>(cmi[j]);
                    const int epi = M->epi;
                    if (epi == EPI_F32) { if (M->res) resv[j] = M->res[(size_t) tok*M->r_tok + row[j]]; }
                    else if (epi != EPI_F16) {     // RoPE (mode NORM): theta by the reference's float recurrence (ggml_rope_cache_init), from the per-forward table when there is one
                        const int ip = (row[j] % L.rope.head_dim) >> 1;
                        if (L.rope.tab) cs[j] = *(const float2 *)(L.rope.tab + ((size_t) tok * (L.rope.head_dim >> 1) + ip) * 2);
                        else cs[j] = mx_rope_cs((float) L.rope.pos[tok], ip, L.rope.theta_scale, L.rope.freq_scale, L.rope.attn_factor);
                    }
                }
            }
        }
        __syncthreads();
        if (first) mx_stamp<STAMP>(stp, 4, lane);
#pragma unroll
        for (int j = 0; j < MX_GMAX; ++j) {
            if (j < nb) {
                const f32x4 * rb = red + (size_t)(par*G + j) * NM * NW * MX_RLD;
                float s[NM][2];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const float2 x = *(const float2 *)((const float *)(rb + (m*NW + w)*MX_RLD + n + 16*h) + 2*pr);
                    s[m][0] = row_sum_f(x.x); s[m][1] = row_sum_f(x.y);
                }
                const mx_matp M = mx_mat<KOFF>(cmi[j]);
                const int epi = DUAL ? EPI_F32 : M->epi;
                const float v0 = (w & 1) ? s[0][1] : s[0][0];
                char * const optr = M->out + (size_t) row[j]*M->o_row;
                if (DUAL) {
                    const float v1 = (w & 1) ? s[NM-1][1] : s[NM-1][0];
                    if (act) *(float *)(optr + (size_t) tok*M->o_tok) = (v0 / (1.0f + expf(-v0))) * v1;
                } else if (epi == EPI_F32) {
                    if (act) {
                        float o = M->res ? v0 + resv[j] : v0; if (M->relu) o = o > 0.f ? o : 0.f;
                        if (!M->ids) *(float *)(optr + (size_t) tok*M->o_tok) = o;
                        else for (int i = 0; i < M->n_ids; ++i) if (M->ids[i] == tok) *(float *)(optr + (size_t) i*M->o_tok) = o;      // output rows that select this token
                    }
                } else if (epi == EPI_F16) {
                    if (act) *(__half *)(optr + (size_t) tok*M->o_tok) = __float2half_rn(v0);
                } else {   // RoPE on the row pair (2p, 2p+1) = neighbouring DPP rows = lanes l, l ^ 16
                    const float prt = __shfl_xor(v0, 16);
                    if (act) {
                        const float x0 = (row[j] & 1) ? prt : v0, x1 = (row[j] & 1) ? v0 : prt;
                        const float y = (row[j] & 1) ? x0*cs[j].y + x1*cs[j].x : x0*cs[j].x - x1*cs[j].y;
                        if (epi == EPI_ROPE_F32) *(float *)(optr + (size_t) tok*M->o_tok) = y;
                        else                     *(__half *)(optr + (size_t) tok*M->o_tok) = __float2half_rn(y);
                    }
                }
            }
        }
        if (nbuf >= 2) par ^= 1; else __syncthreads();          // single buffer: the next batch parks into the slots just read
        if (first) mx_stamp<STAMP>(stp, 5, lane);
        first = false; cgi += nb;
    };
    const int ngr = bid < total ? (total - bid + nblk - 1) / nblk : 0;
    mq_aop<TYPE> P;                 // HOIST: this wave's single unit of activations (k <= 4096), read from the image once
    if constexpr (HOIST) { if (nu > 0) P.load(A, wave, lane); }
    int cslot = 0;
    for (int g0 = 0; g0 < ngr; g0 += G) {
        const int nb = ngr - g0 < G ? ngr - g0 : G;
        for (int j = 0; j < nb; ++j) {
            for (int cu = 0; cu < nu; ++cu) {         // (a wave without units, k < 4096, parks zeros)
                const int unit = DUAL ? wave + (cu >> 1)*NW : wave + cu*NW;
                mx_wait_vm((issued - consumed - 1) * mx_dma_ops<TILE>() + ph_ops);   // what was requested behind this tile may stay in flight
                mq_tfrag<TYPE> f;
                f.template load<false>(myring + (size_t) cslot * TILE, lane, 0);
                if constexpr (HOIST) {
                    if (DUAL && (cu & 1)) mq_run_pre<TYPE>(f, P, lane, acc[NM - 1]); else mq_run_pre<TYPE>(f, P, lane, acc[0]);
                } else {
                    if (DUAL && (cu & 1)) mq_proc<TYPE, 1>::run(f, A, unit, lane, acc[NM - 1]); else mq_proc<TYPE, 1>::run(f, A, unit, lane, acc[0]);
                }
                ++consumed;
                if (lgrp < total) request_next(); else phantom_into(cslot);        // into the slot just read (ring order = stream order)
                cslot = cslot + 1 == D ? 0 : cslot + 1;
            }
            park(j);
        }
        reduce_batch(nb);
    }
    mx_stamp<STAMP>(stp, 6, lane);
}

// HOIST (k <= 4096 and a K-quant): the activation operands of a wave's single unit stay in registers across the row groups
template <int TYPE, bool DUAL, bool HOIST, bool STAMP = false>
__global__ void __launch_bounds__(16*WAVE) k_mmx(const mmvq_launch L, const int T, const int flags, unsigned long long * stamps, const mx_next nx) {
    mmx_body<TYPE, DUAL, HOIST, STAMP, 0>(L, T, flags, blockIdx.x, gridDim.x, stamps, nx);
}
// Two launches that read the same activations but hold weights of different types (Q4_K_M: wq | wk are Q4_K, wv is Q6_K in half of the
// layers) as ONE grid: blocks [0, gridA) run launch A's body, the rest launch B's
template <int TA, int TB, bool HOIST>
__global__ void __launch_bounds__(16*WAVE) k_mmx2(const mmvq_launch LA, const mmvq_launch LB, const int T, const int flagsA, const int flagsB, const int gridA, const mx_next nx) {
    if ((int) blockIdx.x < gridA) mmx_body<TA, false, HOIST, false, 0>(LA, T, flagsA, blockIdx.x, gridA, nullptr, nx);
    else                          mmx_body<TB, false, HOIST, false, (int) sizeof(mmvq_launch)>(LB, T, flagsB, blockIdx.x - gridA, gridDim.x - gridA, nullptr, nx);
}
