// kernels_mmt.hip -- quantised weight x int8 activation products on the matrix cores, reading the TILED weight layout
// (tile_layout.h): the mat-vec of the EAGLE draft step (1 token), of tree verification (2..8 tokens) and, in groups of 8
// tokens, of prompts.  Arithmetic and lane maps are those of kernels_mmq.hip (shared code: mmq_device.h) -- the CPU backend's
// Q8_K / Q8_0 activation quantisation and integer dot products (R/ggml/src/ggml-quants.c:194-215,2479-2512,
// R/ggml/src/ggml-cpu/ggml-cpu-quants.c) -- what changes is how the bytes move:
//   * weights: a wave-instruction fetches 1 KiB of contiguous, 16-byte aligned HBM (a tile piece), instead of 16 rows x 64 B
//     at row stride; nothing is shifted or swapped between lanes for alignment;
//   * activations: with T * ceil(k/4096) <= 8 the block quantises them itself (wave w owns super-blocks w, w+16, ..; the
//     RMS-norm row sums meet once in LDS), while its first weight tiles are already in flight: the stand-alone quantiser launch
//     (4 per transformer layer, 15 % of kernel time in round 1) is gone for wq|wk|wv, wo, gate|up, the LM head and all
//     single-token products; bigger images (ffn_down at 6 tokens, prompts) still come from k_quant_q8K through HBM scratch;
//   * the folded RMS_NORM [* w] result is written out by block 0 as a side effect, so a reader outside the graph view the
//     plugin was handed still finds it;
//   * split-K: the 16 waves of a block split the units of a 16-row group; partial tiles meet in LDS (double-buffered: one
//     barrier per group) and 512 threads reduce + run the epilogue (16 lanes per output quad, DPP row sums) instead of wave 0.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "mmq_device.h"
#include "tile_layout.h"
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define MT_RLD 65                       // f32x4 slots per wave in a reduction buffer (64 lanes + 1: spreads the readers over the banks)

__device__ __forceinline__ float half_row_sum_f(float v) { v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HMIR>(v); return v; }   // 8 lanes


// Diagnostic build of the kernel (STAMP = true, only ever launched with GGML_MI355X_MMT_STAMPS set): lane 0 of every wave writes the 100 MHz
// s_memrealtime clock at the phase boundaries into a debug ring (4 launches x 256 blocks x 16 waves x 8 stamps) that nothing else reads;
// scripts/mmt_stamps.py turns it into the per-phase breakdown under profiles/.  The product instantiations contain no stamp code.
#define MMT_NSTAMP 12
__device__ unsigned long long * g_mmt_stamps = nullptr;
template <bool STAMP> __device__ __forceinline__ void mmt_stamp(unsigned long long * base, int idx, int lane) {
    if constexpr (STAMP) { if (lane == 0 && base) base[idx] = __builtin_amdgcn_s_memrealtime(); }
}
template <bool STAMP> struct mmt_stamper {      // handed into the quantiser: stamps 7.. (activations landed | partial sums reduced | after the norm barrier | scale known)
    unsigned long long * base; int lane;
    __device__ __forceinline__ void operator()(int idx) const { mmt_stamp<STAMP>(base, idx, lane); }
    __device__ __forceinline__ void landed(int idx) const { if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); mmt_stamp<STAMP>(base, idx, lane); } }
};

// ---- in-kernel activation quantiser (16 waves).  Lanes carry TOKENS: LG = 64, 32, 16 or 8 lanes serve one token (1, 2, <= 4, <= 8 tokens
// per launch), lane = LG t + p, so all the tokens of a super-block are quantised by one pass of straight-line code and the cross-lane
// steps are log2(LG) DPP / permute steps -- no readlane, no per-token loop (the wave-per-(token, super-block) form costs ~450
// instructions per unit on every CU: 8 us per launch at 6 tokens).  Wave w owns super-blocks w, w + 16, ...
//   K-quants (Q8_K rule, quantize_row_q8_K_ref, R/ggml/src/ggml-quants.c:2479-2512): lane p holds the float4s f = p + LG i (i < 64/LG) of
//     the super-block (coalesced runs per token).  The scale comes from the FIRST element of largest magnitude: max(+x) and max(-x)
//     decide its sign at once unless a positive and a negative element tie, which takes the scan path.  q = nearest_int(iscale x) by
//     the reference's own magic-number addition (its MIN(127, .) never binds: |iscale x| <= 127 (1 + 2^-23)).
//   Q8_0 rule (quantize_row_q8_0_ref :194-215): lane p holds 256/LG consecutive elements; LG/8 lanes share a 32-element block.
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <int LG> __device__ __forceinline__ float grp_max_f(float v) {           // max over the LG lanes of a token group, result in every lane
    v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_HMIR>(v));
    if (LG >= 16) v = fmaxf(v, dpp_f<DPP_MIR>(v));
    if (LG >= 32) v = max_xw<16>(v);
    if (LG >= 64) v = max_xw<32>(v);
    return v;
}
template <int LG> __device__ __forceinline__ int grp_min_i(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v)); v = min(v, dpp_i<DPP_XOR2>(v)); v = min(v, dpp_i<DPP_HMIR>(v));
    if (LG >= 16) v = min(v, dpp_i<DPP_MIR>(v));
    if (LG >= 32) v = min_xw<16>(v);
    if (LG >= 64) v = min_xw<32>(v);
    return v;
}
template <int LG> __device__ __forceinline__ double grp_sum_d(double v) {
    int2 p = *(int2 *) &v;
#define DSTEP(C) { int2 q; q.x = dpp_i<C>(p.x); q.y = dpp_i<C>(p.y); v += *(double *) &q; p = *(int2 *) &v; }
    DSTEP(DPP_XOR1) DSTEP(DPP_XOR2) DSTEP(DPP_HMIR)
    if (LG >= 16) DSTEP(DPP_MIR)
#undef DSTEP
    if (LG >= 32) v = sum_xw<16>(v);
    if (LG >= 64) v = sum_xw<32>(v);
    return v;
}
// max of three on one instruction (fmaxf chains go through NaN canonicalisation: 4x the instructions); operands are finite activations
__device__ __forceinline__ float max3f(float a, float b, float c)  { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3nf(float a, float b, float c) { float r; asm("v_max3_f32 %0, -%1, -%2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3af(float a, float b, float c) { float r; asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ const float * act_ptr(const act_src & a, int t, int e) {
    return (a.X2 && e >= a.ksplit) ? a.X2 + (size_t) t*a.xs2 + (e - a.ksplit) : a.X + (size_t) t*a.xs + e;
}
// low bytes of four dwords -> one dword
__device__ __forceinline__ int pack_b0(const i32x4 b) {
    return (int)(__builtin_amdgcn_perm((uint32_t) b.y, (uint32_t) b.x, 0x0c0c0400u) | __builtin_amdgcn_perm((uint32_t) b.w, (uint32_t) b.z, 0x04000c0cu));
}
// element offset of float4 i of lane p inside the super-block
template <bool Q80, int LG> __device__ __forceinline__ constexpr int mt_eoff(int i) { return Q80 ? 4*i : 4*LG*i; }
template <bool Q80, int LG> __device__ __forceinline__ int mt_poff(int p) { return Q80 ? (256/LG)*p : 4*p; }

// cross-lane integer maxima over the LG lanes of a token group (result in every lane of the group)
template <int LG> __device__ __forceinline__ int grp_max_i(int v) {
    v = max(v, dpp_i<DPP_XOR1>(v)); v = max(v, dpp_i<DPP_XOR2>(v)); v = max(v, dpp_i<DPP_HMIR>(v));
    if (LG >= 16) v = max(v, dpp_i<DPP_MIR>(v));
    if (LG >= 32) { uint32_t a, b; lane_pair<16>((uint32_t) v, a, b); v = max((int) a, (int) b); }
    if (LG >= 64) { uint32_t a, b; lane_pair<32>((uint32_t) v, a, b); v = max((int) a, (int) b); }
    return v;
}
template <int LG> __device__ __forceinline__ uint32_t grp_max_u(uint32_t v) {
    v = max(v, (uint32_t) dpp_i<DPP_XOR1>((int) v)); v = max(v, (uint32_t) dpp_i<DPP_XOR2>((int) v)); v = max(v, (uint32_t) dpp_i<DPP_HMIR>((int) v));
    if (LG >= 16) v = max(v, (uint32_t) dpp_i<DPP_MIR>((int) v));
    if (LG >= 32) { uint32_t a, b; lane_pair<16>(v, a, b); v = max(a, b); }
    if (LG >= 64) { uint32_t a, b; lane_pair<32>(v, a, b); v = max(a, b); }
    return v;
}
// One super-block of every token slot of the wave -> int8 image, scales, split block sums.  Round 3 (the one positive result of the mat-vec
// lab, profiles/r03_matvec_lab_findings.md: -5..7 % on the launches without a norm): the signed extrema are tracked on the INTEGER pipe -- a
// non-negative float orders like its bit pattern, among negative floats the unsigned order is the magnitude order: v_max3_i32 / v_max3_u32, no NaN
// canonicalisation, single DPP-fused cross-lane steps -- and the block sums are not reduced across lanes (24 DPP adds per super-block) but read
// back: the wave's own int8 bytes, 32 consecutive per lane, summed with v_dot4 by one lane per (token, 32-group).
template <int TYPE, int LG>
__device__ __forceinline__ void mt_quant_sb(const f32x4v (&x)[64/LG], const bool tv, const int t, const int p, const int T, const int sb, const int ldq, const int nsb, int8_t * lq, float * ldy, char * lrec) {
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int NF = 64/LG;
    int8_t * dst = lq + (size_t) t*ldq + sb*256 + mt_poff<Q80, LG>(p);
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
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const int q0 = (int) roundf(x[i].x*id), q1 = (int) roundf(x[i].y*id), q2 = (int) roundf(x[i].z*id), q3 = (int) roundf(x[i].w*id);
            const int pk = (q0 & 0xff) | ((q1 & 0xff) << 8) | ((q2 & 0xff) << 16) | (q3 << 24);
            if (tv) *(int *)(dst + 4*i) = pk;
        }
        if (tv && (p % (LG/8)) == 0) ldy[t*nsb + sb*8 + p / (LG/8)] = __half2float(__float2half_rn(dd));
    } else {
        // Q8_K rule (quantize_row_q8_K_ref): the scale comes from the FIRST element of largest magnitude.  pmb = bits of max(+x) (signed integer
        // order, floor +0), nmb = bits of the most negative element (unsigned order, floor -0)
        int pmb = 0; uint32_t nmb = 0x80000000u;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            const i32x4 b = __builtin_bit_cast(i32x4, x[i]);
            pmb = max(max(b.x, b.y), pmb); pmb = max(max(b.z, b.w), pmb);
            nmb = max(max((uint32_t) b.x, (uint32_t) b.y), nmb); nmb = max(max((uint32_t) b.z, (uint32_t) b.w), nmb);
        }
        pmb = grp_max_i<LG>(pmb); nmb = grp_max_u<LG>(nmb);
        const int nmag = (int)(nmb & 0x7fffffffu);
        const float pm = __int_as_float(pmb), nm = __int_as_float(nmag);
        const float amax = pmb > nmag ? pm : nm;
        float mx = pmb > nmag ? pm : -nm;
        const bool tie = pmb == nmag && pmb != 0;
        if (__any(tie)) {      // a positive and a negative element share the largest magnitude: the first one in element order decides
            int key = 0x7fffffff;
#pragma unroll
            for (int i = NF - 1; i >= 0; --i) {
                const float xe[4] = { x[i].x, x[i].y, x[i].z, x[i].w };
#pragma unroll
                for (int e = 3; e >= 0; --e) if (fabsf(xe[e]) == amax) key = ((4*(p + LG*i) + e) << 1) | (xe[e] < 0.f ? 1 : 0);
            }
            key = grp_min_i<LG>(key);
            if (tie) mx = (key & 1) ? -amax : amax;
        }
        const bool nz = pmb != 0 || nmag != 0;
        const float iscale = nz ? -127.f / mx : 0.f;
        const float dd = nz ? 1.0f / iscale : 0.f;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
            f32x4v r;
            {   // rounded product, THEN nearest_int's magic addition (ggml-quants.c:559-565): the two roundings must not be contracted into an FMA
#pragma clang fp contract(off)
                const f32x4v m = x[i] * iscale;
                r = m + 12582912.f;
            }
            const int pk = pack_b0(__builtin_bit_cast(i32x4, r));
            if (tv) *(int *)(dst + 4*LG*i) = pk;
        }
        if (tv && p == 0) ldy[t*nsb + sb] = dd;
        // block sums for the mins / offset MFMA, split as s = 128 h + l (mmvq_device.h): the wave reads its own bytes back (LDS is in order per
        // wave), lane (token tt, 32-group g) sums 32 consecutive int8 with dot4; 8 * 64/LG lanes take part
        const int lane = t*LG + p, tt = lane >> 3, g = lane & 7;
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
template <int TYPE, int LG>
__device__ __forceinline__ void mt_load_sb(const act_src & a, f32x4v (&x)[64/LG], const bool tv, const int t, const int p, const int sb) {
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    // one base pointer per (token, super-block) -- CONCAT sources switch at a super-block boundary -- then constant offsets; the lanes past
    // the last token are masked off: the vector memory pipe is paid per active lane (64 B / clock / CU), idle token slots are not free
    if (tv) {
        const float * base = act_ptr(a, t, sb*256) + mt_poff<Q80, LG>(p);
#pragma unroll
        for (int i = 0; i < 64/LG; ++i) x[i] = *(const f32x4v *)(base + mt_eoff<Q80, LG>(i));
    } else {
#pragma unroll
        for (int i = 0; i < 64/LG; ++i) x[i] = (f32x4v){ 0.f, 0.f, 0.f, 0.f };
    }
}
template <int TYPE, int LG, int NSB, class PFN, class STP>
__device__ __forceinline__ void mt_norm_quant(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                              int8_t * lq, float * ldy, char * lrec, double * rd, const int lane, const int wave, PFN prefetch, const int pfpos, STP stp) {
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int NF = 64/LG;
    const int t = lane / LG, p = lane % LG;
    const bool tv = t < T;
    f32x4v xv[NSB][NF], wraw;
    double ss = 0.0;
#pragma unroll
    for (int c = 0; c < NSB; ++c) if (wave + 16*c < nun) mt_load_sb<TYPE, LG>(a, xv[c], tv, t, p, wave + 16*c);
    // norm weights of this wave's super-block (one super-block per wave: k <= 4096, the usual case): ONE coalesced 1 KiB request, lane l
    // taking floats 4l..4l+3, handed to the token groups through the wave's 1 KiB of LDS behind the partial sums -- every token group
    // fetching its own copy would be LG-fold traffic through the vector memory pipe, in the prologue's critical path
    const bool wlds = NSB == 1 && a.norm_w && wave < nun;
    f32x4v * wst = (f32x4v *)((char *) rd + 1024) + wave*64;
    if (wlds) wraw = *(const f32x4v *)(a.norm_w + wave*256 + 4*lane);
    stp.landed(7);
    if (pfpos == 0) prefetch();
#pragma unroll
    for (int c = 0; c < NSB; ++c) {
        if (wave + 16*c < nun) {
#pragma unroll
            for (int i = 0; i < NF; ++i) { const f32x4v q = xv[c][i] * xv[c][i]; ss += (double) q.x; ss += (double) q.y; ss += (double) q.z; ss += (double) q.w; }
        }
    }
    ss = grp_sum_d<LG>(ss);
    if (p == 0 && tv) rd[t*16 + wave] = ss;
    if (wlds) wst[lane] = wraw;
    stp(8);
    if (pfpos == 1) prefetch();
    __syncthreads();
    stp(9);
    // the 16 wave partials of this lane's token: spread over the first 16 / 8 lanes of the group, then the same group sum
    double tot = 0.0;
    if (tv) {
        if (LG >= 16) { if (p < 16) tot = rd[t*16 + p]; }
        else tot = rd[t*16 + p] + rd[t*16 + p + 8];
    }
    tot = grp_sum_d<LG>(tot);
    // sum / k: an exponent step when k is a power of two (4096, 8192: bit-identical to the division, a fraction of its instruction chain)
    const float mean = (k & (k - 1)) == 0 ? (float) __builtin_ldexp(tot, -__builtin_ctz(k)) : (float)(tot / (double) k);
    const float s1 = 1.0f / sqrtf(mean + a.eps);
    stp(10);
    if (pfpos >= 2) prefetch();
#pragma unroll
    for (int c = 0; c < NSB; ++c) {
        const int sb = wave + 16*c;
        if (sb < nun) {
#pragma unroll
            for (int i = 0; i < NF; ++i) {
                const int fi = Q80 ? NF*p + i : p + LG*i;                    // float4 of the super-block this register holds
                const int e = sb*256 + 4*fi;
                f32x4v v = xv[c][i] * s1;
                if (a.norm_w) v *= (NSB == 1) ? wst[fi] : *(const f32x4v *)(a.norm_w + e);
                if (a.norm_out && blockIdx.x == 0 && tv) *(f32x4v *)(a.norm_out + (size_t) t*a.norm_os + e) = v;
                xv[c][i] = v;
            }
            mt_quant_sb<TYPE, LG>(xv[c], tv, t, p, T, sb, ldq, nsb, lq, ldy, lrec);
        }
    }
}
// `prefetch` issues the block's first weight-tile loads.  It runs right AFTER the first activation loads: vmcnt retires in issue order, so
// activations requested behind the (HBM-cold) weight tiles would wait for them; this way the L2-warm activations come back first and
// the quantiser's arithmetic overlaps the weight latency.
template <int TYPE, int LG, class PFN, class STP>
__device__ __forceinline__ void mt_quantise_lg(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                               int8_t * lq, float * ldy, char * lrec, double * rd /* LDS [8][16] + 16 KiB */, const int lane, const int wave, PFN prefetch, const int pfpos, STP stp) {
    constexpr int NF = 64/LG;
    const int t = lane / LG, p = lane % LG;
    const bool tv = t < T;
    if (a.norm) {
        // ggml_compute_forward_rms_norm_f32 (R/ggml/src/ggml-cpu/ggml-cpu.c:7098-7144): sum of x*x (float products) in double,
        // mean = (float)(sum / k), scale = 1/sqrtf(mean + eps); then MUL by the norm weight.  k <= 8192 here: two super-blocks per wave at
        // most; with one (k <= 4096, the usual case) the norm weights are requested together with the activations, ahead of the barrier.
        if (nun <= 16) mt_norm_quant<TYPE, LG, 1>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
        else           mt_norm_quant<TYPE, LG, 2>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
    } else {
        // (keeping a second super-block of activations in flight per wave measured +-0 at k = 11008 and costs registers: not done)
        f32x4v x[NF];
        if (wave < nun) mt_load_sb<TYPE, LG>(a, x, tv, t, p, wave);
        stp.landed(7);
        if (pfpos == 0) prefetch();
        for (int sb = wave; sb < nun; sb += 16) {
            if (sb != wave) mt_load_sb<TYPE, LG>(a, x, tv, t, p, sb);
            mt_quant_sb<TYPE, LG>(x, tv, t, p, T, sb, ldq, nsb, lq, ldy, lrec);
            if (pfpos == 1 && sb == wave) prefetch();
        }
        if (pfpos >= 2 || (pfpos == 1 && wave >= nun)) prefetch();
    }
}
template <int TYPE, class PFN, class STP>
__device__ __forceinline__ void mt_quantise(const act_src & a, const int T, const int k, const int nun, const int nsb, const int ldq,
                                            int8_t * lq, float * ldy, char * lrec, double * rd, const int lane, const int wave, PFN prefetch, const int pfpos, STP stp) {
    if (T == 1)      mt_quantise_lg<TYPE, 64>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
    else if (T == 2) mt_quantise_lg<TYPE, 32>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
    else if (T <= 4) mt_quantise_lg<TYPE, 16>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
    else             mt_quantise_lg<TYPE, 8>(a, T, k, nun, nsb, ldq, lq, ldy, lrec, rd, lane, wave, prefetch, pfpos, stp);
}

static inline size_t mmt_img_bytes(int T, int k, bool q80) {
    const size_t nsc = q80 ? k/32 : k/256;
    return (size_t) T*(k + 16) + (((size_t) T*nsc*4 + 15) & ~(size_t) 15) + (q80 ? 0 : (size_t) T*nsc*32);
}
static inline int mmt_nw(int T) { return T <= 8 ? 16 : 8; }             // one token group: 16 waves (128 VGPRs); 2-3 groups need 256 VGPRs: 8 waves, two blocks per CU
static inline size_t mmt_lds_bytes(int T, int k, bool dual, bool q80, int nbuf) {
    const size_t red = (size_t) nbuf * (dual ? 2 : 1) * mmt_nw(T) * MT_RLD * 16;
    return mmt_img_bytes(T, k, q80) + (red > 17408 ? red : 17408);      // the quantiser's scratch shares the reduction buffers: 1 KiB of partial sums + 16 KiB of norm weights
}

// the kernel body: block `bid` of `nblk` (k_mmt: the whole grid; k_mmt2: one of the two partitions of a mixed-type launch)
template <int TYPE, bool DUAL, bool PF, int TG, int NW, bool STAMP>
__device__ __forceinline__ void mmt_body(const mmvq_launch & L, const int T, const int nbuf_knob, const int launch_id, const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nbuf = nbuf_knob & 0xff, pfpos = nbuf_knob >> 8;      // pfpos: where the prologue issues the first weight tiles (GGML_MI355X_MMT_PFPOS)
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int NM = DUAL ? 2 : 1;
    constexpr int TILE = mq_tfrag<TYPE>::TILE;
    const int k = L.k, nun = k/256, nsb = Q80 ? k/32 : nun, ldq = k + 16;
    int8_t * lq  = (int8_t *) smem;
    float  * ldy = (float *)(smem + (size_t) T*ldq);
    char   * lrec = (char *) ldy + (((size_t) T*nsb*4 + 15) & ~(size_t) 15);
    f32x4  * red = (f32x4 *)(lrec + (Q80 ? 0 : (size_t) T*nsb*32));
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    unsigned long long * stp = nullptr;
    if constexpr (STAMP) { if (g_mmt_stamps && blockIdx.x < 256) stp = g_mmt_stamps + ((size_t)((launch_id & 3) * 256 + blockIdx.x) * 16 + wave) * MMT_NSTAMP; }
    mmt_stamp<STAMP>(stp, 0, lane);

    const int c0 = L.m[0].rows >> 4;
    const int c1 = (!DUAL && L.n_mat > 1) ? L.m[1].rows >> 4 : 0;
    const int c2 = (!DUAL && L.n_mat > 2) ? L.m[2].rows >> 4 : 0;
    const int total = c0 + c1 + c2;
    const int nu = wave < nun ? ((nun - wave + NW - 1) / NW) * NM : 0;     // (matrix, unit) steps of this wave per row group

    const char * tp[NM];                                                     // tile row of the current group: unit u at tp + u*TILE
    auto set_rows = [&](int g, int & mi, int & row0) {
        if (g < c0) { mi = 0; row0 = g*16; } else if (g < c0 + c1) { mi = 1; row0 = (g - c0)*16; } else { mi = 2; row0 = (g - c0 - c1)*16; }
        tp[0] = L.m[mi].W + (size_t) row0 * L.m[mi].row_bytes;
        if (DUAL) tp[NM - 1] = L.m[1].W + (size_t) row0 * L.m[1].row_bytes;
    };
    auto unit_of  = [&](int u) -> int { return DUAL ? wave + (u >> 1)*NW : wave + u*NW; };
    auto unit_ptr = [&](int u) -> const char * { return (DUAL ? tp[u & 1] : tp[0]) + (size_t) unit_of(u) * TILE; };

    mq_tfrag<TYPE> fa, fb;          // fb unused without PF
    int grp = bid, mi = 0, row0 = 0;
    if (grp < total) set_rows(grp, mi, row0);
    auto prefetch = [&]() {          // the first tiles are in flight across the prologue
        if (grp < total) {
            if (nu > 0) fa.load(unit_ptr(0), lane, 0);
            if (PF && nu > 1) fb.load(unit_ptr(1), lane, 0);
        }
    };

    if (L.act.pre || NW != 16) {   // image written by the quantiser launch (HBM scratch) -> LDS; token rows padded by 16 bytes against bank conflicts
        prefetch();
        const int nthr = NW*WAVE, n16row = k/16;
        const i32x4 * src = (const i32x4 *) L.act.pre;
        for (int c = tid; c < T*n16row; c += nthr) { const int t = c / n16row, o = c - t*n16row; *(i32x4 *)(lq + (size_t) t*ldq + o*16) = src[c]; }
        const float * sd = (const float *)(L.act.pre + (size_t) T*k);
        for (int c = tid; c < T*nsb; c += nthr) ldy[c] = sd[c];
        if (!Q80) {
            const char * sr = L.act.pre + act_img_bytes(true, T, k) + (TYPE == GGML_TYPE_Q6_K ? (size_t) T*nsb*32 : 0);     // 4-byte aligned only
            for (int c = tid; c < T*nsb*2; c += nthr) ((i32x4 *) lrec)[c] = ld16(sr + (size_t) c*16);
        }
    } else if constexpr (NW == 16) {
        mt_quantise<TYPE>(L.act, T, k, nun, nsb, ldq, lq, ldy, lrec, (double *) red, lane, wave, prefetch, pfpos, mmt_stamper<STAMP>{ stp, lane });
    }
    mmt_stamp<STAMP>(stp, 1, lane);                                 // this wave's share of the activation image is in LDS
    __syncthreads();
    mmt_stamp<STAMP>(stp, 2, lane);
    const mq_act A = { lq, ldq, ldy, lrec, nsb, T };

    int par = 0; bool first = true;
    while (grp < total) {
        float acc[NM][TG][4];
#pragma unroll
        for (int m = 0; m < NM; ++m)
#pragma unroll
            for (int t = 0; t < TG; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[m][t][r] = 0.f;
        if (PF) {
            for (int u = 0; u < nu; u += 2) {
                mq_proc<TYPE, TG>::run(fa, A, unit_of(u), lane, acc[0]);                                   // DUAL: even steps = gate
                if (u + 2 < nu) fa.load(unit_ptr(u + 2), lane, 0);
                if (u + 1 >= nu) break;
                mq_proc<TYPE, TG>::run(fb, A, unit_of(u + 1), lane, acc[NM - 1]);                          // DUAL: odd steps = up
                if (u + 3 < nu) fb.load(unit_ptr(u + 3), lane, 0);
            }
        } else {
            for (int u = 0; u < nu; u += NM) {
                mq_proc<TYPE, TG>::run(fa, A, unit_of(u), lane, acc[0]);
                if (DUAL) { fa.load(unit_ptr(u + 1), lane, 0); mq_proc<TYPE, TG>::run(fa, A, unit_of(u), lane, acc[NM - 1]); }
                if (u + NM < nu) fa.load(unit_ptr(u + NM), lane, 0);
            }
        }
        if (first) mmt_stamp<STAMP>(stp, 3, lane);                 // first group's units done (weights arrived + MFMA work)
        // next group's first tiles go out before this group's reduction / epilogue
        const int cmi = mi, crow0 = row0, gn = grp + nblk;
        if (gn < total) {
            set_rows(gn, mi, row0);
            if (nu > 0) fa.load(unit_ptr(0), lane, 0);
            if (PF && nu > 1) fb.load(unit_ptr(1), lane, 0);
        }
        // ---- split-K reduction: every wave parks its partial tile, then 16 lanes per output quad (row n, tokens 4h..4h+3) add the
        // 16 waves x 2 classes in a fixed order and the quad's first four lanes finish one token each
#pragma unroll
        for (int tgi = 0; tgi < TG; ++tgi) {
            if (tgi*8 >= T) break;
            f32x4 * rb = red + (size_t) par * NM * NW * MT_RLD;
            if (nbuf < 2 || TG > 1) __syncthreads();                       // single buffer: the readers of the previous tile must be done
#pragma unroll
            for (int m = 0; m < NM; ++m) { const f32x4 v = { acc[m][tgi][0], acc[m][tgi][1], acc[m][tgi][2], acc[m][tgi][3] }; rb[(m*NW + wave)*MT_RLD + lane] = v; }
            __syncthreads();
            if (first) mmt_stamp<STAMP>(stp, 4, lane);
            if (tid < 32*NW) {
                const int w = tid % NW, q = tid / NW, n = q & 15, h = q >> 4;
                float s[NM][4];
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    const f32x4 x = rb[(m*NW + w)*MT_RLD + q] + rb[(m*NW + w)*MT_RLD + q + 32];
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[m][r] = NW == 16 ? row_sum_f(x[r]) : half_row_sum_f(x[r]);
                }
                const mmvq_mat & M = L.m[cmi];
                const int row = crow0 + n, tok = 8*tgi + 4*h + (w & 3);
                const bool act = w < 4 && tok < T;
                float v0 = (w & 3) == 0 ? s[0][0] : (w & 3) == 1 ? s[0][1] : (w & 3) == 2 ? s[0][2] : s[0][3];
                if (DUAL) {
                    const float v1 = (w & 3) == 0 ? s[NM-1][0] : (w & 3) == 1 ? s[NM-1][1] : (w & 3) == 2 ? s[NM-1][2] : s[NM-1][3];
                    if (act) *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = (v0 / (1.0f + expf(-v0))) * v1;
                } else if (M.epi == EPI_F32) {
                    if (act) {
                        float o = v0; if (M.res) o += M.res[(size_t) tok*M.r_tok + row]; if (M.relu) o = o > 0.f ? o : 0.f;
                        if (!M.ids) *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = o;
                        else for (int j = 0; j < M.n_ids; ++j) if (M.ids[j] == tok) *(float *)(M.out + (size_t) row*M.o_row + (size_t) j*M.o_tok) = o;      // output rows that select this token
                    }
                } else if (M.epi == EPI_F16) {
                    if (act) *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(v0);
                } else {   // RoPE (mode NORM) on the row pair (2p, 2p+1) = quads q, q^1 = lanes l, l^NW; theta by the reference's float recurrence (ggml_rope_cache_init)
                    const float pr = __shfl_xor(v0, NW);
                    if (act) {
                        const int ip = (row % L.rope.head_dim) >> 1;
                        float c, sn;
                        if (L.rope.tab) { const float2 cs = *(const float2 *)(L.rope.tab + ((size_t) tok * (L.rope.head_dim >> 1) + ip) * 2); c = cs.x; sn = cs.y; }     // same numbers, computed once per forward pass
                        else {
                            float th = (float) L.rope.pos[tok];
                            for (int j = 0; j < ip; ++j) th *= L.rope.theta_scale;
                            const float a = L.rope.freq_scale * th;
                            c = cosf(a) * L.rope.attn_factor; sn = sinf(a) * L.rope.attn_factor;
                        }
                        const float x0 = (row & 1) ? pr : v0, x1 = (row & 1) ? v0 : pr;
                        const float y = (row & 1) ? x0*sn + x1*c : x0*c - x1*sn;
                        if (M.epi == EPI_ROPE_F32) *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = y;
                        else                       *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(y);
                    }
                }
            }
            if (nbuf >= 2 && TG == 1) par ^= 1;
        }
        if (first) mmt_stamp<STAMP>(stp, 5, lane);
        first = false;
        grp = gn;
    }
    mmt_stamp<STAMP>(stp, 6, lane);
}
template <int TYPE, bool DUAL, bool PF, int TG, int NW, bool STAMP = false>
__global__ void __launch_bounds__(NW*WAVE) k_mmt(const mmvq_launch L, const int T, const int nbuf, const int launch_id) {
    mmt_body<TYPE, DUAL, PF, TG, NW, STAMP>(L, T, nbuf, launch_id, blockIdx.x, gridDim.x);
}
// Two launches that read the same activations but hold weights of different types (Q4_K_M: wq | wk are Q4_K, wv is Q6_K in half of the
// layers) as ONE grid: blocks [0, gridA) run launch A's body, the rest launch B's -- each block builds only its own type's activation
// image, registers are the maximum of the two bodies, and a transformer layer stays at five launches instead of six.
template <int TA, int TB, bool PFA, bool PFB>
__global__ void __launch_bounds__(16*WAVE) k_mmt2(const mmvq_launch LA, const mmvq_launch LB, const int T, const int nbuf, const int gridA) {
    if ((int) blockIdx.x < gridA) mmt_body<TA, false, PFA, 1, 16, false>(LA, T, nbuf, 0, blockIdx.x, gridA);
    else                          mmt_body<TB, false, PFB, 1, 16, false>(LB, T, nbuf, 0, blockIdx.x - gridA, gridDim.x - gridA);
}

// theta of pair ip at position p by the reference's float recurrence (ggml_rope_cache_init: theta starts at p and is multiplied by theta_scale
// per pair), then cos / sin scaled by attn_factor: exactly what the fused RoPE epilogue computes per element -- here once per (token, pair)
__global__ void __launch_bounds__(128) k_rope_table(const int32_t * __restrict__ pos, float * __restrict__ tab, const int half, const float theta_scale, const float freq_scale, const float attn_factor) {
    const int tok = blockIdx.x;
    for (int ip = threadIdx.x; ip < half; ip += blockDim.x) {
        float th = (float) pos[tok];
        for (int j = 0; j < ip; ++j) th *= theta_scale;
        const float a = freq_scale * th;
        *(float2 *)(tab + ((size_t) tok * half + ip) * 2) = make_float2(cosf(a) * attn_factor, sinf(a) * attn_factor);
    }
}
void mi_rope_table(hipStream_t st, const int32_t * pos, int T, int head_dim, float theta_scale, float freq_scale, float attn_factor, float * tab) {
    k_rope_table<<<T, 128, 0, st>>>(pos, tab, head_dim / 2, theta_scale, freq_scale, attn_factor);
}

// ---------------------------------------------------------------- host side
static int device_cus() {
    static const int n = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256; return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }();
    return n;
}
static void ensure_attr(const void * fn) { mi_allow_big_lds(fn); }
#ifdef MI_LAB          // the STAMP = true instantiations and their plumbing exist in the lab build only (build.py --lab)
static unsigned long long * g_stamp_host_ptr = nullptr;
static bool mmt_stamps_on() {
    static const bool on = [] {
        if (!mi_lab_env("GGML_MI355X_MMT_STAMPS")) return false;
        const size_t n = (size_t) 4 * 256 * 16 * MMT_NSTAMP * 8;
        HIP_CHECK(hipMalloc((void **) &g_stamp_host_ptr, n)); HIP_CHECK(hipMemset(g_stamp_host_ptr, 0, n));
        HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_mmt_stamps), &g_stamp_host_ptr, sizeof(void *)));
        return true;
    }();
    return on;
}
// diagnostic: copies the stamp ring (4 x 256 x 16 x 8 u64) to `out`; returns the number of u64 written (0 when stamps are off)
extern "C" __attribute__((visibility("default"))) int ggml_backend_mi355x_mmt_stamps(unsigned long long * out) {
    if (!g_stamp_host_ptr) return 0;
    const size_t n = (size_t) 4 * 256 * 16 * MMT_NSTAMP;
    HIP_CHECK(hipDeviceSynchronize());
    HIP_CHECK(hipMemcpy(out, g_stamp_host_ptr, n * 8, hipMemcpyDeviceToHost));
    return (int) n;
}
#endif
static bool mmt_dual_pf() { static const bool v = mi_lab_env("GGML_MI355X_MMT_DUAL_PF") != nullptr; return v; }      // A/B: double-buffered fragments in the gate|up launch
static int mmt_pfpos() { static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMT_PFPOS"); return e ? atoi(e) : 2; }(); return v; }   // 0 behind the activation loads | 1 behind the partial sums | 2 once the norm scale is known
static int mmt_nbuf() { static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMT_NBUF"); return e ? atoi(e) : 2; }(); return v; }

template <int TYPE, bool DUAL, bool PF, int TG> static void mmt_launch_one(hipStream_t st, int T, const mmvq_launch & L) {
    constexpr int NW = TG == 1 ? 16 : 8;
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    int nbuf = (TG == 1 && mmt_nbuf() >= 2 && mmt_lds_bytes(T, L.k, DUAL, Q80, 2) <= 160*1024) ? 2 : 1;
    const size_t lds = mmt_lds_bytes(T, L.k, DUAL, Q80, nbuf);
    MI_ASSERT(lds <= 160*1024);
    int total = 0;
    if (DUAL) total = L.m[0].rows / 16;
    else for (int i = 0; i < L.n_mat; ++i) total += L.m[i].rows / 16;
    if (total < 1) return;
#ifdef MI_LAB
    if constexpr (TG == 1) {
        if (mmt_stamps_on()) {
            static int launch_id = 0;                                    // (lab runs are single-threaded)
            auto sfn = k_mmt<TYPE, DUAL, PF, TG, NW, true>;
            ensure_attr((const void *) sfn);
            const int sgrid = total < device_cus() ? total : device_cus();
            sfn<<<sgrid, NW*WAVE, lds, st>>>(L, T, nbuf | (mmt_pfpos() << 8), launch_id++);
            return;
        }
    }
#endif
    auto fn = k_mmt<TYPE, DUAL, PF, TG, NW>;
    ensure_attr((const void *) fn);
    const int slots = device_cus() * (NW == 16 ? 1 : ((2*lds <= 160*1024) ? 2 : 1));
    const int grid = total < slots ? total : slots;
    const int pi = mi_prof_begin(st, L, T, DUAL);
    fn<<<grid, NW*WAVE, lds, st>>>(L, T, nbuf | (mmt_pfpos() << 8), 0);
    mi_prof_end(st, pi);
}
template <int TYPE> static void mmt_launch_type(hipStream_t st, int T, const mmvq_launch & L) {
    constexpr bool PF = TYPE != GGML_TYPE_Q6_K && TYPE != GGML_TYPE_Q8_0 && TYPE != GGML_TYPE_Q4_0;      // wide fragments: single-buffered (128 VGPRs = 4 waves/SIMD)
    const int tg = (T + 7) / 8;
    MI_ASSERT(tg >= 1 && tg <= (L.swiglu ? 2 : 3));
    if (L.swiglu) { if (tg == 1) { if (PF && mmt_dual_pf()) mmt_launch_one<TYPE, true, true, 1>(st, T, L); else mmt_launch_one<TYPE, true, false, 1>(st, T, L); } else mmt_launch_one<TYPE, true, false, 2>(st, T, L); }
    else if (tg == 1) mmt_launch_one<TYPE, false, PF, 1>(st, T, L);
    else if (tg == 2) mmt_launch_one<TYPE, false, PF, 2>(st, T, L);
    else              mmt_launch_one<TYPE, false, PF, 3>(st, T, L);
}
static void mmt_launch(hipStream_t st, int type, int T, const mmvq_launch & L) {
    switch (type) {
        case GGML_TYPE_Q4_K: mmt_launch_type<GGML_TYPE_Q4_K>(st, T, L); break;
        case GGML_TYPE_Q5_K: mmt_launch_type<GGML_TYPE_Q5_K>(st, T, L); break;
        case GGML_TYPE_Q6_K: mmt_launch_type<GGML_TYPE_Q6_K>(st, T, L); break;
        case GGML_TYPE_Q8_0: mmt_launch_type<GGML_TYPE_Q8_0>(st, T, L); break;
        case GGML_TYPE_Q4_0: mmt_launch_type<GGML_TYPE_Q4_0>(st, T, L); break;
        default: MI_ABORT("mmt: unsupported weight type %d", type);
    }
}
static bool mmt_fits(int type, int T, int k, bool swiglu) {
    const bool q80 = type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q4_0;
    const int tcap = type == GGML_TYPE_Q4_0 ? (swiglu ? 8 : 16) : (swiglu ? 16 : 24);      // register budget of the token-group variants
    return T >= 1 && T <= tcap && mmt_lds_bytes(T, k, swiglu, q80, 1) <= 158*1024;
}
// tokens one pass takes: a multiple of 8 (whole token groups) above 8
static int mmt_max_tokens(int type, int k, bool swiglu) {
    for (int t = 24; t >= 8; t -= 8) if (mmt_fits(type, t, k, swiglu)) return t;
    int t = 7;
    while (t > 0 && !mmt_fits(type, t, k, swiglu)) --t;
    return t;
}
static bool mmt_inline_quant(int T, const mmvq_launch & L) {
    static const bool off = mi_lab_env("GGML_MI355X_MMT_NO_INLINE") != nullptr;
    static const int maxsb = [] { const char * e = mi_lab_env("GGML_MI355X_MMT_INLINE_MAXSB"); return e ? atoi(e) : 3; }();   // super-blocks per wave the in-kernel quantiser may take
    if (off || T > 8) return false;
    if (L.act.X2 && (L.act.ksplit % 256)) return false;
    const int per_wave = (L.k/256 + 15) / 16;
    return L.act.norm ? per_wave <= 2 : per_wave <= maxsb;
}


// ---------------------------------------------------------------- big batches: one pass over the weights
// Reference: ggml_cuda_op_mul_mat_cublas (R/ggml/src/ggml-cuda/ggml-cuda.cu:1164-1225: dequantise the whole matrix to fp16, then a BLAS GEMM)
// and mul_mat_q (mmq.cuh:2590) for 9..63 tokens.  Here the arithmetic stays the CPU backend's (int8 activations, integer dots), so the
// matrix-core mat-vec is re-tiled as a GEMM instead: a block owns 64 weight rows x 32 tokens, its 16 waves are 4 row groups x 4 token
// groups (a wave = 16 rows x 8 tokens, the accumulator of the mat-vec kernel), and k is walked unit by unit with the activation tile of
// the step (32 tokens x 256 int8 + scales + block sums, ~10 KB) double-buffered in LDS from the quantised image in HBM scratch.  Per step a
// block moves 9 KB of weights (HBM; the four token quarters of a row block run on one XCD and share them in L2) against 10 KB of
// activations (L2): balanced, every CU busy, weights streamed once for the whole batch instead of once per 24 tokens.
#define BB_LDQ 272                      // bytes per token row of an activation tile (256 + 16: conflict-free 16-byte operand reads)
// TGW = 8-token groups one wave multiplies with a dequantised tile (1, 2 or 4): the block keeps its 64 rows x 32 tokens and runs
// 16 / TGW waves, every weight tile being unpacked 4 / TGW times per block instead of four.  Measured on the 128-token 7B prompt
// (scripts/prompt_time.py, profiles/r02_ab_second_half.txt): 12.7 ms at TGW = 1, 14.9 at 2, 21.7 at 4 -- the kernel lives on its 16
// waves hiding LDS operand and MFMA latency, not on unpacking work; TGW = 1 is what runs (GGML_MI355X_MMT_BB_TGW selects the others)
template <int TYPE, int TGW>
__global__ void __launch_bounds__(1024 / TGW) k_mmt_bb(const mmvq_launch L, const int T, const int n_rq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int TILE = mq_tfrag<TYPE>::TILE;
    constexpr int NT = 1024 / TGW, NTG = 4 / TGW;                    // threads; token groups of waves
    constexpr int DN = Q80 ? 8 : 1;                                  // activation scales per (token, unit)
    constexpr int BUF = 32*BB_LDQ + 32*DN*4 + (Q80 ? 0 : 32*32);     // one tile: quants | scales | split block sums
    constexpr int NITEM = 512 + 32*DN + (Q80 ? 0 : 64);              // staging items of a tile: 16-byte quant pieces | scales | 16-byte record pieces
    constexpr int NI = (NITEM + NT - 1) / NT;
    const int k = L.k, nun = k/256, nsb_img = Q80 ? k/32 : nun;
    const int tid = threadIdx.x, lane = tid % WAVE, wave = tid / WAVE;
    // block -> (row quad, token quarter): the quarters of one row quad get block ids 8 apart (one XCD under round-robin placement: speed only)
    const int b = blockIdx.x, nq = (T + 31) / 32;
    const int rq = (b & 7) + 8 * ((b >> 3) / nq), tq = (b >> 3) % nq;
    if (rq >= n_rq) return;
    const int rg = wave / NTG, tg = wave % NTG;
    const mmvq_mat & M = L.m[0];
    const int row0 = rq*64 + rg*16, t0 = tq*32;
    const bool rows_ok = row0 < M.rows;                              // the last row quad may hold fewer than four groups
    const char * tp = M.W + (size_t)(rows_ok ? row0 : 0) * M.row_bytes;
    // image (mi_quant_act): [T][k] int8 | [T][nsb] f32 d | ([T][k/16] i16 bsums) | rec32 [T][nsb][32] | rec16 [T][nsb][32]
    const char * img = L.act.pre;
    const float * img_d = (const float *)(img + (size_t) T*k);
    const char * img_rec = Q80 ? nullptr : img + act_img_bytes(true, T, k) + (TYPE == GGML_TYPE_Q6_K ? (size_t) T*nsb_img*32 : 0);
    // a step = two units (one barrier per 512 k): the step's two activation tiles sit side by side in one LDS buffer
    i32x4 rq4[2][NI];
    auto load_tile = [&](int step) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int u = 2*step + j;
            if (u >= nun) break;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = tid + i*NT;
                i32x4 v = {0, 0, 0, 0};
                if (c < 512) { const int tok = t0 + (c >> 4); if (tok < T) v = *(const i32x4 *)(img + (size_t) tok*k + u*256 + 16*(c & 15)); }
                else if (c < 512 + 32*DN) { const int tok = t0 + (c - 512) / DN; if (tok < T) v.x = __float_as_int(img_d[(size_t) tok*nsb_img + u*DN + (c - 512) % DN]); }
                else if (!Q80 && c < NITEM) { const int tok = t0 + ((c - 512 - 32*DN) >> 1); if (tok < T) v = ld16(img_rec + ((size_t) tok*nsb_img + u)*32 + 16*((c - 512 - 32*DN) & 1)); }
                rq4[j][i] = v;
            }
        }
    };
    auto store_tile = [&](char * buf0, int step) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (2*step + j >= nun) break;
            char * buf = buf0 + j*BUF;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int c = tid + i*NT;
                if (c < 512) *(i32x4 *)(buf + (c >> 4)*BB_LDQ + 16*(c & 15)) = rq4[j][i];
                else if (c < 512 + 32*DN) ((int *)(buf + 32*BB_LDQ))[c - 512] = rq4[j][i].x;
                else if (!Q80 && c < NITEM) *(i32x4 *)(buf + 32*BB_LDQ + 32*DN*4 + (c - 512 - 32*DN)*16) = rq4[j][i];
            }
        }
    };
    mq_tfrag<TYPE> fa, fb;
    load_tile(0);
    fa.template load<false>(tp, lane, 0);
    if (nun > 1) fb.template load<false>(tp + TILE, lane, 0);
    store_tile(smem, 0);
    __syncthreads();
    float acc[TGW][4];
#pragma unroll
    for (int t = 0; t < TGW; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][r] = 0.f;
    const int tok_w = t0 + 8*TGW*tg;                                // first token of this wave
    const bool mine = rows_ok && tok_w < T;
    const int nsteps = (nun + 1) / 2;
    for (int st = 0; st < nsteps; ++st) {
        char * cur = smem + (st & 1)*2*BUF;
        if (st + 1 < nsteps) load_tile(st + 1);
        {
            const int u = 2*st;
            const mq_act A = { (const int8_t *) cur + 8*TGW*tg*BB_LDQ, BB_LDQ, (const float *)(cur + 32*BB_LDQ) + 8*TGW*tg*DN, cur + 32*BB_LDQ + 32*DN*4 + 8*TGW*tg*32, DN, T - tok_w };
            if (mine) mq_proc<TYPE, TGW>::run(fa, A, 0, lane, acc);
            if (u + 2 < nun) fa.template load<false>(tp + (size_t)(u + 2)*TILE, lane, 0);
        }
        if (2*st + 1 < nun) {
            const int u = 2*st + 1; const char * c1 = cur + BUF;
            const mq_act A = { (const int8_t *) c1 + 8*TGW*tg*BB_LDQ, BB_LDQ, (const float *)(c1 + 32*BB_LDQ) + 8*TGW*tg*DN, c1 + 32*BB_LDQ + 32*DN*4 + 8*TGW*tg*32, DN, T - tok_w };
            if (mine) mq_proc<TYPE, TGW>::run(fb, A, 0, lane, acc);
            if (u + 2 < nun) fb.template load<false>(tp + (size_t)(u + 2)*TILE, lane, 0);
        }
        if (st + 1 < nsteps) store_tile(smem + ((st + 1) & 1)*2*BUF, st + 1);
        __syncthreads();
    }
    // classes (lanes l, l ^ 32) meet in registers; lanes kq < 2 hold row n, tokens tok_w + 8 t + 4 kq + r
    const int n = lane & 15, kq = lane >> 4, row = row0 + n;
#pragma unroll
    for (int t = 0; t < TGW; ++t) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sum_xw<32>(acc[t][r]);
        if (mine && kq < 2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = tok_w + 8*t + 4*kq + r;
                if (tok < T) { float o = v[r]; if (M.res) o += M.res[(size_t) tok*M.r_tok + row]; if (M.relu) o = o > 0.f ? o : 0.f; *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = o; }
            }
        }
    }
}
static int mmt_bb_tgw() { static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMT_BB_TGW"); const int t = e ? atoi(e) : 1; return (t == 1 || t == 2 || t == 4) ? t : 1; }(); return v; }
template <int TYPE> static void mmt_bb_launch(hipStream_t st, int T, const mmvq_launch & L) {
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;
    constexpr int DN = Q80 ? 8 : 1;
    const size_t lds = 4 * (size_t)(32*BB_LDQ + 32*DN*4 + (Q80 ? 0 : 32*32));      // two buffers of two unit tiles
    MI_ASSERT(L.act.pre && L.n_mat == 1 && L.m[0].rows % 16 == 0);
    const int n_rq = (L.m[0].rows + 63) / 64, nq = (T + 31) / 32;
    const int grid = ((n_rq + 7) / 8) * 8 * nq;
    const int pi = mi_prof_begin(st, L, T, false);
    switch (mmt_bb_tgw()) {
        case 2:  k_mmt_bb<TYPE, 2><<<grid, 512, lds, st>>>(L, T, n_rq); break;
        case 4:  k_mmt_bb<TYPE, 4><<<grid, 256, lds, st>>>(L, T, n_rq); break;
        default: k_mmt_bb<TYPE, 1><<<grid, 1024, lds, st>>>(L, T, n_rq); break;
    }
    mi_prof_end(st, pi);
}
static void mmt_bb_dispatch(hipStream_t st, int type, int T, const mmvq_launch & L) {
    switch (type) {
        case GGML_TYPE_Q4_K: mmt_bb_launch<GGML_TYPE_Q4_K>(st, T, L); break;
        case GGML_TYPE_Q5_K: mmt_bb_launch<GGML_TYPE_Q5_K>(st, T, L); break;
        case GGML_TYPE_Q6_K: mmt_bb_launch<GGML_TYPE_Q6_K>(st, T, L); break;
        case GGML_TYPE_Q8_0: mmt_bb_launch<GGML_TYPE_Q8_0>(st, T, L); break;
        case GGML_TYPE_Q4_0: mmt_bb_launch<GGML_TYPE_Q4_0>(st, T, L); break;
        default: MI_ABORT("mmt_bb: unsupported weight type %d", type);
    }
}
static int mmt_bb_min_tokens() { static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMT_BB_MIN_T"); return e ? atoi(e) : 25; }(); return v; }
// image of a whole big batch (all of k): a ring of 4 x 8 MiB slots next to the small-image ring, keyed like it
#define MI_BIG_SLOTS 4
#define MI_BIG_SLOT_BYTES ((size_t) 8 << 20)
static const char * mmt_big_image(hipStream_t st, int type, int T, int t0, const mmvq_launch & L, mi_act_cache * cache, const void * key) {
    if (!cache->big_pool) HIP_CHECK(hipMalloc((void **) &cache->big_pool, MI_BIG_SLOTS * MI_BIG_SLOT_BYTES));
    const int kq = mi_traits(type).blck == 256;
    int hit = -1;
    if (key) for (int i = 0; i < MI_BIG_SLOTS; ++i) { const auto & e = cache->big[i]; if (e.key == key && e.epoch == cache->epoch && e.t0 == t0 && e.T == T && e.kq == kq && e.k == L.k) { hit = i; break; } }
    if (hit < 0) {
        hit = cache->big_next; cache->big_next = (cache->big_next + 1) % MI_BIG_SLOTS;
        mi_quant_act(st, type, T, L.act, L.k, cache->big_pool + (size_t) hit * MI_BIG_SLOT_BYTES);
        cache->big[hit] = { key, cache->epoch, t0, T, kq, L.k };
    }
    return cache->big_pool + (size_t) hit * MI_BIG_SLOT_BYTES;
}

// quantised image of tokens [t0, t0+T) x k-range of L in an HBM scratch slot (shared between launches reading the same activations)
static const char * mmt_image(hipStream_t st, int type, int T, int t0, const mmvq_launch & L, mi_act_cache * cache, const void * key) {
    const int kq = mi_traits(type).blck == 256;
    int hit = -1;
    if (key) for (int i = 0; i < MI_ACT_SLOTS; ++i) { const auto & e = cache->e[i]; if (e.key == key && e.epoch == cache->epoch && e.t0 == t0 && e.T == T && e.kq == kq && e.k == L.k) { hit = i; break; } }
    if (hit < 0) {
        hit = cache->next; cache->next = (cache->next + 1) % MI_ACT_SLOTS;
        mi_quant_act(st, type, T, L.act, L.k, cache->pool + (size_t) hit * cache->slot_bytes);
        cache->e[hit] = { key, cache->epoch, t0, T, kq, L.k };
    }
    return cache->pool + (size_t) hit * cache->slot_bytes;
}

static bool mmt_pair_off() { static const bool v = mi_lab_env("GGML_MI355X_MMT_NO_PAIR") != nullptr; return v; }
bool mi_mmt_pair_supported(int typeA, int typeB, int T, const mmvq_launch & LA) {
    if (mmt_pair_off() || T < 1 || T > 8 || typeB != GGML_TYPE_Q6_K || !(typeA == GGML_TYPE_Q4_K || typeA == GGML_TYPE_Q5_K)) return false;
    if (!mmt_inline_quant(T, LA)) return false;                       // both bodies quantise in their prologue (same activations, same k)
    return mmt_lds_bytes(T, LA.k, false, false, 1) <= 158*1024;
}
template <int TA> static void mmt_pair_launch(hipStream_t st, int T, const mmvq_launch & LA, const mmvq_launch & LB) {
    const int nbuf = (mmt_nbuf() >= 2 && mmt_lds_bytes(T, LA.k, false, false, 2) <= 160*1024) ? 2 : 1;
    const size_t lds = mmt_lds_bytes(T, LA.k, false, false, nbuf);
    double bA = 0, bB = 0; int gA = 0, gB = 0;
    for (int i = 0; i < LA.n_mat; ++i) { bA += (double) LA.m[i].rows * LA.m[i].row_bytes; gA += LA.m[i].rows / 16; }
    for (int i = 0; i < LB.n_mat; ++i) { bB += (double) LB.m[i].rows * LB.m[i].row_bytes; gB += LB.m[i].rows / 16; }
    // CUs in proportion to the bytes of the two partitions (a split that equalises whole rounds of row groups instead measured 1 % slower:
    // profiles/r02_ab_second_half.txt)
    const int cus = device_cus();
    int gridA = (int)(cus * bA / (bA + bB) + 0.5); if (gridA < 1) gridA = 1; if (gridA > cus - 1) gridA = cus - 1; if (gridA > gA) gridA = gA;
    int gridB = cus - gridA; if (gridB > gB) gridB = gB;
    auto fn = k_mmt2<TA, GGML_TYPE_Q6_K, true, false>;
    ensure_attr((const void *) fn);
    const int pa = mi_prof_begin(st, LA, T, false);                    // accounted as one launch: the bytes of both
    fn<<<gridA + gridB, 16*WAVE, lds, st>>>(LA, LB, T, nbuf | (mmt_pfpos() << 8), gridA);
    mi_prof_end(st, pa);
    mi_prof_add_bytes(mi_launch_bytes(LB, T, false) - (double) T * LB.k * 4);      // B's weights and outputs; the activations were counted once
}
void mi_mmt_run_pair(hipStream_t st, int typeA, int typeB, int T, const mmvq_launch & LA, const mmvq_launch & LB) {
    MI_ASSERT(typeB == GGML_TYPE_Q6_K && LA.tiled && LB.tiled && LA.k == LB.k && !LA.swiglu && !LB.swiglu && !LA.act.pre && !LB.act.pre);
    if (typeA == GGML_TYPE_Q4_K) mmt_pair_launch<GGML_TYPE_Q4_K>(st, T, LA, LB); else mmt_pair_launch<GGML_TYPE_Q5_K>(st, T, LA, LB);
}
void mi_mmt_run(hipStream_t st, int type, int Ttot, const mmvq_launch & L0, mi_act_cache * cache, const void * key) {
    MI_ASSERT(L0.tiled && cache && cache->pool && L0.k % 256 == 0);
    const int tile = mi_tile_bytes(type);
    const bool swiglu = L0.swiglu != 0;
    int tmax = mmt_max_tokens(type, L0.k, swiglu);
    while (tmax > 8 && mi_act_image_bytes(type, tmax, L0.k) > cache->slot_bytes) tmax -= 8;
    // k too long for the LDS image at the wanted tokens per pass (ffn_down at > 8 tokens, k = 28672 at >= 5 tokens): k-chunks of whole
    // units, every later chunk adding to the output through the residual input.  Plain outputs only (the row norm needs the whole row).
    const int Tw = Ttot >= 24 ? 24 : (Ttot >= 16 ? 16 : (Ttot > 8 ? 8 : Ttot));
    // the one-pass big-batch kernel: one matrix, any epilogue but the row selection; RoPE epilogues need the table (graph.cpp builds it for
    // batches of any size); the image's quantiser launch takes a folded norm / SwiGLU product
    const bool rope_epi = L0.n_mat == 1 && (L0.m[0].epi == EPI_ROPE_F32 || L0.m[0].epi == EPI_ROPE_F16);
    const bool plain_bb = L0.n_mat == 1 && !swiglu && !L0.m[0].relu && !L0.m[0].ids && !L0.act.X2 && (!rope_epi || L0.rope.tab) && mi_bb_supported(type);
    const bool old_ok = L0.n_mat == 1 && !swiglu && L0.m[0].epi == EPI_F32 && !L0.m[0].relu && !L0.m[0].ids && !L0.act.X2;      // round 2's kernel (Q4_0): plain f32 outputs only
    const bool plain = old_ok && !L0.act.norm && !L0.act.G;
    MI_ASSERT(!L0.act.G || ((plain_bb || old_ok) && Ttot >= mmt_bb_min_tokens()));       // silu(G) * X exists in the quantiser launches only (graph.cpp: can_defer_swiglu)
    if ((plain_bb || old_ok) && Ttot >= mmt_bb_min_tokens()) {
        // one pass over the weights for the whole batch (a3); token passes only when the batch's int8 image outgrows a scratch slot
        int tpass = Ttot;
        while (tpass > 32 && mi_act_image_bytes(type, tpass, L0.k) > MI_BIG_SLOT_BYTES) tpass = (tpass / 2 + 31) / 32 * 32;
        for (int t0 = 0; t0 < Ttot; t0 += tpass) {
            const int T = (Ttot - t0) < tpass ? (Ttot - t0) : tpass;
            mmvq_launch L = L0;
            L.act.X += (size_t) t0 * L.act.xs;
            if (L.act.G) L.act.G += (size_t) t0 * L.act.gs;
            if (L.act.norm_out) L.act.norm_out += (size_t) t0 * L.act.norm_os;
            L.m[0].out += (size_t) t0 * L.m[0].o_tok;
            if (L.m[0].res) L.m[0].res += (size_t) t0 * L.m[0].r_tok;
            if (L.rope.tab) L.rope.tab += (size_t) t0 * (L.rope.head_dim >> 1) * 2;
            L.act.pre = mmt_big_image(st, type, T, t0, L, cache, key);
            if (plain_bb) mi_bb_run(st, type, T, L); else mmt_bb_dispatch(st, type, T, L);      // (Q4_0, GGML_MI355X_BB_OLD=1: round 2's kernel)
        }
        return;
    }
    if (tmax < Tw && plain) {
        const int nun = L0.k / 256;
        int nch = 2, kc = 0;
        for (; nch <= 16; ++nch) { kc = (nun + nch - 1) / nch * 256; if (mmt_fits(type, Tw, kc, false) && mi_act_image_bytes(type, Tw, kc) <= cache->slot_bytes) break; }
        if (nch <= 16) {
            for (int t0 = 0; t0 < Ttot; t0 += Tw) {
                const int T = (Ttot - t0) < Tw ? (Ttot - t0) : Tw;
                for (int k0 = 0, c = 0; k0 < L0.k; k0 += kc, ++c) {
                    mmvq_launch L = L0;
                    L.k = (L0.k - k0) < kc ? (L0.k - k0) : kc;
                    L.act.X += (size_t) t0 * L.act.xs + k0; L.act.pre = nullptr;
                    L.m[0].W += (size_t)(k0 / 256) * tile;                                  // same row group, later units
                    L.m[0].out += (size_t) t0 * L.m[0].o_tok;
                    if (c == 0) { if (L.m[0].res) L.m[0].res += (size_t) t0 * L.m[0].r_tok; }
                    else { L.m[0].res = (const float *) L.m[0].out; L.m[0].r_tok = L.m[0].o_tok / 4; }
                    if (!mmt_inline_quant(T, L)) L.act.pre = mmt_image(st, type, T, t0, L, cache, key ? (const void *)((const char *) key + k0 + 1) : nullptr);
                    mmt_launch(st, type, T, L);
                }
            }
            return;
        }
    }
    MI_ASSERT(tmax >= 1);
    for (int i = 0; i < L0.n_mat; ++i) MI_ASSERT(!L0.m[i].ids || Ttot <= tmax);      // row selection indexes whole-launch tokens: one pass only
    for (int t0 = 0; t0 < Ttot; t0 += tmax) {
        const int T = (Ttot - t0) < tmax ? (Ttot - t0) : tmax;
        mmvq_launch L = L0;
        L.act.X += (size_t) t0 * L.act.xs;
        if (L.act.X2) L.act.X2 += (size_t) t0 * L.act.xs2;
        if (L.act.norm_out) L.act.norm_out += (size_t) t0 * L.act.norm_os;
        L.act.pre = nullptr;
        for (int i = 0; i < L.n_mat; ++i) { L.m[i].out += (size_t) t0 * L.m[i].o_tok; if (L.m[i].res) L.m[i].res += (size_t) t0 * L.m[i].r_tok; }
        if (L.rope.pos) L.rope.pos += t0;
        if (L.rope.tab) L.rope.tab += (size_t) t0 * (L.rope.head_dim >> 1) * 2;
        if (!mmt_inline_quant(T, L)) {
            MI_ASSERT(mi_act_image_bytes(type, T, L.k) <= cache->slot_bytes);
            L.act.pre = mmt_image(st, type, T, t0, L, cache, key);
        }
        mmt_launch(st, type, T, L);
    }
}
