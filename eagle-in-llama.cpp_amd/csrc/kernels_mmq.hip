// kernels_mmq.hip -- quantised weight (Q4_K, Q5_K, Q6_K, Q8_0) x int8 activation products for 2..24 tokens per pass on the matrix cores.
//
// Same arithmetic as kernels_mmvq.hip (the CPU backend's: Q8_K activations, the integer dot products of
// ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K, R/ggml/src/ggml-cpu/ggml-cpu-quants.c), different machine mapping: the dp4a
// kernel spends ~20 VALU instructions per (row, token, 32 weights), which makes a 6-token tree verification
// VALU-bound long before HBM is busy.  Here v_mfma_i32_16x16x64_i8 does the integer dots of 16 weight rows against
// all tokens at once:
//     B (N = weight rows)  lane (n = lane&15, kq = lane>>4): 16 unpacked quants of row n in k-slots 16kq..16kq+15
//     A (M = token,class)  lane (i = lane&15, kq): M rows 0..7 = the tokens, M rows 8..15 = the tokens again;
//                          the two classes are non-zero in DIFFERENT k-slot groups, so one MFMA (K = 64) returns the
//                          sums over two different sub-blocks separately -- each needs its own scale
//     C                    lane (n, g = lane>>4), reg r: class g>>1, token 4*(g&1) + r
// After the MFMA a lane holds, for its row, the sub-block sums of four tokens: the sub-block scale is one integer
// multiply-add per (row, token), nothing is reduced across lanes (the two classes meet in the split-K reduction),
// and a wave stores 16 consecutive rows.
//   * Q4_K / Q5_K: a 16-byte load of qs is two B operands (low / high nibbles = sub-blocks 2g / 2g+1); lanes kq<2 carry
//     groups g = 0, 1, lanes kq>=2 groups 2, 3 -> 4 MFMAs per super-block.
//   * Q6_K: 16-element sub-blocks: a B operand holds four of them; two MFMAs with complementary activation masks read it.
//   * mins (Q4_K/Q5_K) and the -32 offset (Q6_K) are sum_j m_j * bsum_j: one more MFMA against the block sums, which
//     the quantiser stores split as 128*h + l (both int8; l = class 0, h = class 1).
//   * Q8_0: two 32-element blocks per MFMA (one per class), fp32 scale d_w * d_a per (row, token, block) as in
//     ggml_vec_dot_q8_0_q8_0.
//   * TG groups of 8 tokens per pass (prompt / wide verification batches): the unpacked B operand meets TG activation operands.
// A 512/1024-thread block owns 16 rows; its waves split the super-blocks of the row (split-K) and reduce through LDS in
// a fixed order.  The quantised activation image (mi_quant_act) is copied into LDS once per block.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "mmvq_device.h"
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>

typedef float f32x4  __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 mfma_i8(const i32x4 a, const i32x4 b) {
    const i32x4 z = {0, 0, 0, 0};
    return __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, z, 0, 0, 0);
}
// 16 bytes from an address that is only 2-byte aligned (Q6_K blocks are 210 bytes).  In this kernel's lane layout (16 rows x 64 B
// per wave instruction) loads that are not dword-aligned run at about half rate (scripts/probe_unaligned.hip: 3.1 vs 5.6 TB/s;
// scripts/probe_pattern.hip: the Q6_K pattern alone 21 us -> 13 us for ffn_down), so a fragment is kept as five aligned dwords
// (the fifth only read when needed -- it never leaves the 210-byte block) and funnel-shifted by 0 or 16 bits at its first use.
struct raw16 { i32x4 v; int e; };
__device__ __forceinline__ raw16 ld16_a2(const char * p) {
    const int mis = (int)((uintptr_t) p & 2);
    const char * q = p - mis;                                  // pointer arithmetic, not an integer round trip: keeps the global address space
    raw16 r; r.v = ld16(q); r.e = 0;
    if (mis) __builtin_memcpy(&r.e, q + 16, 4);
    return r;
}
__device__ __forceinline__ i32x4 fix16(const raw16 & r, int sh) {
    i32x4 o;
    o.x = __builtin_amdgcn_alignbit(r.v.y, r.v.x, sh); o.y = __builtin_amdgcn_alignbit(r.v.z, r.v.y, sh);
    o.z = __builtin_amdgcn_alignbit(r.v.w, r.v.z, sh); o.w = __builtin_amdgcn_alignbit(r.e, r.v.w, sh);
    return o;
}
__device__ __forceinline__ int byte_of(uint32_t v, int j) { return (int)((v >> (8*j)) & 0xffu); }
__device__ __forceinline__ int sbyte_of(uint32_t v, int j) { return (int)(int8_t)((v >> (8*j)) & 0xffu); }

// ---- what a lane (n = lane&15, kq = lane>>4) keeps of one (row, super-block): loads only
template <int TYPE> struct mq_frag;
template <> struct mq_frag<GGML_TYPE_Q4_K> {          // qs bytes [32*gA + 16*(kq&1), +16) for gA = (kq>>1)*2 + {0, 1}
    static constexpr int BLK = 144;
    i32x4 hdr, qs[2];
    __device__ __forceinline__ void load(const char * b, int kq, int) {
        hdr = ld16(b);
        qs[0] = ld16(b + 16 + 64*(kq >> 1) + 16*(kq & 1)); qs[1] = ld16(b + 16 + 64*(kq >> 1) + 32 + 16*(kq & 1));
    }
};
template <> struct mq_frag<GGML_TYPE_Q5_K> {
    static constexpr int BLK = 176;
    i32x4 hdr, qh, qs[2];
    __device__ __forceinline__ void load(const char * b, int kq, int) {
        hdr = ld16(b); qh = ld16(b + 16 + 16*(kq & 1));
        qs[0] = ld16(b + 48 + 64*(kq >> 1) + 16*(kq & 1)); qs[1] = ld16(b + 48 + 64*(kq >> 1) + 32 + 16*(kq & 1));
    }
};
template <> struct mq_frag<GGML_TYPE_Q6_K> {          // kq = 2*qb + lh: ql bytes [64*nn + 32*qb + 16*lh, +16) for nn = 0, 1; qh bytes [16*kq, +16):
    static constexpr int BLK = 210;                    // every byte of the block is requested once (the lanes swap qh pieces in registers)
    raw16 ql[2], qh, sc; int dh, sh;
    __device__ __forceinline__ void load(const char * b, int kq, int) {
        sh = (int)((uintptr_t) b & 2) * 8;             // all pieces sit at multiples of 16 from the block start: one shift for the fragment
#pragma unroll
        for (int nn = 0; nn < 2; ++nn) ql[nn] = ld16_a2(b + 64*nn + 32*(kq >> 1) + 16*(kq & 1));
        qh = ld16_a2(b + 128 + 16*kq);
        sc = ld16_a2(b + 192);
        uint16_t d; __builtin_memcpy(&d, b + 208, 2); dh = d;
    }
};

template <> struct mq_frag<GGML_TYPE_Q8_0> {          // a unit = 8 blocks of 34 bytes (f16 d + 32 int8); MFMA a covers blocks 2a (class 0) and 2a+1 (class 1):
    static constexpr int BLK = 272;                    // lane kq holds the 16 quants [16*(kq&1), +16) of block 2a + (kq>>1); blocks are 2-byte aligned
    raw16 q[4]; int dh[4]; int shs;
    __device__ __forceinline__ void load(const char * b, int kq, int nb_left) {
        shs = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int blk = 2*a + (kq >> 1);
            const char * p = b + (blk < nb_left ? blk : 0)*34;            // past the ragged end: any valid block, its product meets zero activations
            const char * qp = p + 2 + 16*(kq & 1);
            shs |= (int)(((uintptr_t) qp & 2) >> 1) << a;
            q[a] = ld16_a2(qp);
            uint16_t d; __builtin_memcpy(&d, p, 2); dh[a] = d;
        }
    }
};

template <> struct mq_frag<GGML_TYPE_Q4_0> {          // a unit = 8 blocks of 18 bytes (f16 d + 16 bytes of nibbles: low = elements 0..15, high = 16..31);
    static constexpr int BLK = 144;                    // MFMA a covers blocks 2a (class 0) and 2a+1 (class 1); lane kq takes the low (kq&1 = 0) or high nibbles
    raw16 q[4]; int dh[4]; int shs;
    __device__ __forceinline__ void load(const char * b, int kq, int nb_left) {
        shs = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int blk = 2*a + (kq >> 1);
            const char * p = b + (blk < nb_left ? blk : 0)*18;
            shs |= (int)(((uintptr_t)(p + 2) & 2) >> 1) << a;
            q[a] = ld16_a2(p + 2);
            uint16_t d; __builtin_memcpy(&d, p, 2); dh[a] = d;
        }
    }
};

// LDS view of the activation image
struct mq_act { const int8_t * q; int ldq; const float * d; const char * rec; int nsb; int T; };   // nsb: scales per token (super-blocks; 32-blocks for Q8_0)

// ---- one super-block of 16 rows x T tokens.  The M dimension of the MFMA carries (token, class): M rows 0..7 are the
// tokens against the k-slots of one sub-block, M rows 8..15 the same tokens against another sub-block, the activations
// being zero in the k-slots of the other class.  C lane (n, g = lane>>4), reg r: M row 4g + r, i.e. class g>>1, token
// 4*(g&1) + r: acc[r] is this lane's share (its class) of out[row n][token 4*(g&1) + r]; classes are added in the reduction.
// TG groups of 8 tokens per pass: the B operands (unpacked quants) are built once and multiplied with TG activation operands.
template <int TYPE, int TG> struct mq_proc;

template <int TYPE, int TG> __device__ __forceinline__ void mq_process_q45(const mq_frag<TYPE> & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) {
    const int i = lane & 15, kq = lane >> 4, g = kq;
    const uint32_t u0 = f.hdr.y, u1 = f.hdr.z, u2 = f.hdr.w;             // get_scale_min_k4 for all eight sub-blocks (ggml-quants.c:631-638)
    const uint32_t s_lo = u0 & 0x3f3f3f3fu, s_hi = (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u);
    const uint32_t m_lo = u1 & 0x3f3f3f3fu, m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u);
    const float dw = h2f((uint16_t)(f.hdr.x & 0xffff)), mw = h2f((uint16_t)((uint32_t) f.hdr.x >> 16));
    // B: lanes kq<2 hold group gA = {0,1}, lanes kq>=2 group gA + 2; k-slots 16*(kq&1).. of the 32-element sub-block.
    // A: M row i<8 = token i, active in k-slots kq<2 (class 0: sub-blocks 2gA, 2gA+1); M row i>=8 = token i-8, active in kq>=2 (class 1)
    const int  tok_a = i & 7, cls_a = i >> 3;
    const bool cv = cls_a == (kq >> 1);
    const int8_t * arow = A.q + tok_a*A.ldq + sb*256 + 128*cls_a + 16*(kq & 1);
    const uint32_t sw = (g >> 1) ? s_hi : s_lo;                           // scales of this lane's class: sub-blocks 4*cls + 0..3
    int isum[TG][4];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) isum[t][r] = 0;
#pragma unroll
    for (int ga = 0; ga < 2; ++ga) {
        i32x4 blo = f.qs[ga] & 0x0F0F0F0F, bhi = (f.qs[ga] >> 4) & 0x0F0F0F0F;
        if constexpr (TYPE == GGML_TYPE_Q5_K) {                           // bit 2g' / 2g'+1 of qh, g' = 2*(kq>>1) + ga
            const i32x4 hb = (kq >> 1) ? (f.qh >> 4) : f.qh;
            blo |= ((hb >> (2*ga)) & 0x01010101) << 4; bhi |= ((hb >> (2*ga + 1)) & 0x01010101) << 4;
        }
        const int s0 = byte_of(sw, 2*ga), s1 = byte_of(sw, 2*ga + 1);
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const bool av = cv && tok_a + 8*t < A.T;
            const i32x4 alo = av ? *(const i32x4 *)(arow + 8*t*A.ldq + 64*ga) : (i32x4)(0);
            const i32x4 ahi = av ? *(const i32x4 *)(arow + 8*t*A.ldq + 64*ga + 32) : (i32x4)(0);
            const i32x4 c0 = mfma_i8(alo, blo);
            const i32x4 c1 = mfma_i8(ahi, bhi);
#pragma unroll
            for (int r = 0; r < 4; ++r) isum[t][r] += __mul24(s0, c0[r]) + __mul24(s1, c1[r]);
        }
    }
    // mins: sum_j m_j * bsum32_j with the sums split as 128*h + l: class 0 = l parts, class 1 = h parts, both in k-slots kq = 0
    i32x4 bm = {0, 0, 0, 0};
    if (kq == 0) { bm.x = (int) m_lo; bm.y = (int) m_hi; }
    const float mscale = (g >> 1) ? 128.f : 1.f;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const bool mv = kq == 0 && tok_a + 8*t < A.T;
        const i32x4 am = mv ? *(const i32x4 *)(A.rec + ((tok_a + 8*t)*A.nsb + sb)*32 + 16*cls_a) : (i32x4)(0);
        const i32x4 cm = mfma_i8(am, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = 8*t + 4*(g & 1) + r;
            const float dy = tok < A.T ? A.d[tok*A.nsb + sb] : 0.f;
            acc[t][r] += (dw*dy)*(float) isum[t][r] - ((mw*dy)*mscale)*(float) cm[r];
        }
    }
}
template <int TG> struct mq_proc<GGML_TYPE_Q4_K, TG> { static __device__ __forceinline__ void run(const mq_frag<GGML_TYPE_Q4_K> & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) { mq_process_q45<GGML_TYPE_Q4_K, TG>(f, A, sb, lane, acc); } };
template <int TG> struct mq_proc<GGML_TYPE_Q5_K, TG> { static __device__ __forceinline__ void run(const mq_frag<GGML_TYPE_Q5_K> & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) { mq_process_q45<GGML_TYPE_Q5_K, TG>(f, A, sb, lane, acc); } };

template <int TG> struct mq_proc<GGML_TYPE_Q6_K, TG> { static __device__ __forceinline__ void run(const mq_frag<GGML_TYPE_Q6_K> & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) {
    const int i = lane & 15, kq = lane >> 4, g = kq, qb = kq >> 1, lh = kq & 1;
    // element 128nn + 32q + l (l = 16lh + byte): ql[64nn + 32(q&1) + l] nibble q>>1, qh[32nn + l] bits 2q, 2q+1; 16-element
    // sub-block s = 8nn + 2q + lh.  B operand (nn, nib): lane kq holds q = qb + 2nib.  Two MFMAs use it: pass p activates
    // the k-slots of the lanes with qb == p; in a pass, class (M rows 0..7 | 8..15) = lh.
    const int  tok_a = i & 7, cls_a = i >> 3;
    const bool cv = cls_a == lh;
    const int8_t * arow = A.q + tok_a*A.ldq + sb*256 + 16*lh;
    int isum[TG][4];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) isum[t][r] = 0;
    const int cls = g >> 1;
    const i32x4 qh_own = fix16(f.qh, f.sh), scv = fix16(f.sc, f.sh);
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        // this lane needs qh bytes [32nn + 16lh, +16): they were loaded by lane group kq' = 2nn + lh of the same row
        const int src = 4*((lane & 15) + 16*(2*nn + lh));
        i32x4 qhn;
        qhn.x = __builtin_amdgcn_ds_bpermute(src, qh_own.x); qhn.y = __builtin_amdgcn_ds_bpermute(src, qh_own.y);
        qhn.z = __builtin_amdgcn_ds_bpermute(src, qh_own.z); qhn.w = __builtin_amdgcn_ds_bpermute(src, qh_own.w);
        const i32x4 hq = qhn >> (2*qb);                                    // bits 2q.. for q = qb (nib 0) at bit 0, for q = qb + 2 (nib 1) at bit 4
        const i32x4 qln = fix16(f.ql[nn], f.sh);
#pragma unroll
        for (int nib = 0; nib < 2; ++nib) {
            const i32x4 b = nib ? (((qln >> 4) & 0x0F0F0F0F) | (hq & 0x30303030)) : ((qln & 0x0F0F0F0F) | ((hq << 4) & 0x30303030));
            // scales[8nn + 4nib + 2p + lh]: the class of a C lane is lh -> shift once, then bytes 0 / 2 are passes p = 0 / 1
            const uint32_t sw = (uint32_t) scv[2*nn + nib] >> (8*cls);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int q = p + 2*nib;
                const int sc = sbyte_of(sw, 2*p);
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    const i32x4 a = (cv && qb == p && tok_a + 8*t < A.T) ? *(const i32x4 *)(arow + 8*t*A.ldq + 128*nn + 32*q) : (i32x4)(0);
                    const i32x4 c = mfma_i8(a, b);
#pragma unroll
                    for (int r = 0; r < 4; ++r) isum[t][r] += __mul24(sc, c[r]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);                                  // keep the two halves apart: bounds the live A / B operands
    }
    // -32 offset: 32 * sum_j scale_j * bsum16_j (class 0 = l parts, class 1 = h parts of the split sums)
    const i32x4 bm = kq == 0 ? scv : (i32x4)(0);
    const float dw = h2f((uint16_t) f.dh);
    const int mscale = (g >> 1) ? 128*32 : 32;
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        const bool mv = kq == 0 && tok_a + 8*t < A.T;
        const i32x4 am = mv ? *(const i32x4 *)(A.rec + ((tok_a + 8*t)*A.nsb + sb)*32 + 16*cls_a) : (i32x4)(0);
        const i32x4 cm = mfma_i8(am, bm);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = 8*t + 4*(g & 1) + r;
            const float dy = tok < A.T ? A.d[tok*A.nsb + sb] : 0.f;
            acc[t][r] += (dw*dy)*(float)(isum[t][r] - mscale*cm[r]);
        }
    }
} };

template <int TG> struct mq_proc<GGML_TYPE_Q8_0, TG> { static __device__ __forceinline__ void run(const mq_frag<GGML_TYPE_Q8_0> & f, const mq_act & A, int unit, int lane, float (&acc)[TG][4]) {
    // ggml_vec_dot_q8_0_q8_0: sumf += sumi * (d_x * d_y) per 32-element block; activations quantised with quantize_row_q8_0 (image d per block)
    const int i = lane & 15, kq = lane >> 4, g = kq;
    const int tok_a = i & 7, cls_a = i >> 3, cls = g >> 1;
    const bool cv = cls_a == (kq >> 1);
    const int nblk = A.nsb;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const i32x4 b = fix16(f.q[a], ((f.shs >> a) & 1) * 16);
        const int blk_a = unit*8 + 2*a + cls_a, blk_c = unit*8 + 2*a + cls;       // block seen by this lane's A rows / owned by its C values
        const float dw = h2f((uint16_t) f.dh[a]);
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const bool av = cv && tok_a + 8*t < A.T && blk_a < nblk;
            const i32x4 am = av ? *(const i32x4 *)(A.q + (tok_a + 8*t)*A.ldq + blk_a*32 + 16*(kq & 1)) : (i32x4)(0);
            const i32x4 c = mfma_i8(am, b);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 8*t + 4*(g & 1) + r;
                const float dy = (tok < A.T && blk_c < nblk) ? A.d[tok*nblk + blk_c] : 0.f;
                acc[t][r] += (float) c[r] * (dw*dy);
            }
        }
    }
} };

template <int TG> struct mq_proc<GGML_TYPE_Q4_0, TG> { static __device__ __forceinline__ void run(const mq_frag<GGML_TYPE_Q4_0> & f, const mq_act & A, int unit, int lane, float (&acc)[TG][4]) {
    // ggml_vec_dot_q4_0_q8_0: sumi = sum (q - 8) * a per 32-element block, sumf += (sumi * d_x) * d_y.  sum q*a and sum a come from two
    // MFMAs against the same activation operand (the second one with an all-ones B), so no block sums are needed in the image.
    const int i = lane & 15, kq = lane >> 4, g = kq;
    const int tok_a = i & 7, cls_a = i >> 3, cls = g >> 1;
    const bool cv = cls_a == (kq >> 1);
    const int nblk = A.nsb;
    const i32x4 ones = (i32x4)(0x01010101);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const i32x4 raw = fix16(f.q[a], ((f.shs >> a) & 1) * 16);
        const i32x4 b = (kq & 1) ? ((raw >> 4) & 0x0F0F0F0F) : (raw & 0x0F0F0F0F);
        const int blk_a = unit*8 + 2*a + cls_a, blk_c = unit*8 + 2*a + cls;
        const float dw = h2f((uint16_t) f.dh[a]);
#pragma unroll
        for (int t = 0; t < TG; ++t) {
            const bool av = cv && tok_a + 8*t < A.T && blk_a < nblk;
            const i32x4 am = av ? *(const i32x4 *)(A.q + (tok_a + 8*t)*A.ldq + blk_a*32 + 16*(kq & 1)) : (i32x4)(0);
            const i32x4 c = mfma_i8(am, b), sa = mfma_i8(am, ones);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int tok = 8*t + 4*(g & 1) + r;
                const float dy = (tok < A.T && blk_c < nblk) ? A.d[tok*nblk + blk_c] : 0.f;
                acc[t][r] += ((float)(c[r] - 8*sa[r]) * dw) * dy;
            }
        }
    }
} };

static inline size_t mmq_lds_bytes(int T, int k, int NW, bool dual, bool q80 = false) {
    const size_t nsc = q80 ? k/32 : k/256;                   // scales per token
    return (size_t) T*(k + 16) + (((size_t) T*nsc*4 + 15) & ~(size_t) 15) + (q80 ? 0 : (size_t) T*nsc*32) + (size_t) NW*64*16*(dual ? 2 : 1);      // the reduction tiles are re-used per token group
}

// PF: double-buffer the weight fragments (the loads of unit u+1 fly while unit u is computed); without it a wave keeps one
// fragment set and relies on the other waves of its SIMD to cover the load latency (Q6_K, Q8_0, gate|up: the register budget for 4 waves/SIMD)
template <int TYPE, bool DUAL, int NW, bool PF, int TG>
__global__ void __launch_bounds__(NW*WAVE) k_mmq(const mmvq_launch L, const int T) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0;     // 32-element blocks, Q8_0 activation image
    const int k = L.k, nsb = Q80 ? k/32 : k/256, nun = Q80 ? (k/32 + 7)/8 : k/256, ldq = k + 16;      // nsb: scales per token, nun: units per row
    int8_t * lq  = (int8_t *) smem;
    float  * ldy = (float *)(smem + (size_t) T*ldq);
    char   * lrec = (char *) ldy + (((size_t) T*nsb*4 + 15) & ~(size_t) 15);
    f32x4  * red = (f32x4 *)(lrec + (Q80 ? 0 : (size_t) T*nsb*32));
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int kq = lane >> 4;
    constexpr int BLK = mq_frag<TYPE>::BLK;
    constexpr int NM = DUAL ? 2 : 1;

    const int c0 = (L.m[0].rows + 15) / 16;
    const int c1 = (!DUAL && L.n_mat > 1) ? (L.m[1].rows + 15) / 16 : 0;
    const int c2 = (!DUAL && L.n_mat > 2) ? (L.m[2].rows + 15) / 16 : 0;
    const int total = c0 + c1 + c2;
    const int nu = wave < nun ? ((nun - wave + NW - 1) / NW) * NM : 0;    // (matrix, super-block) units of this wave per row group

    const char * rp[NM];
    auto set_rows = [&](int g, int & mi, int & row0) {
        if (g < c0) { mi = 0; row0 = g*16; } else if (g < c0 + c1) { mi = 1; row0 = (g - c0)*16; } else { mi = 2; row0 = (g - c0 - c1)*16; }
        rp[0] = L.m[mi].W + (size_t) min(row0 + (lane & 15), L.m[mi].rows - 1) * L.m[mi].row_bytes;
        if (DUAL) rp[NM - 1] = L.m[1].W + (size_t) min(row0 + (lane & 15), L.m[1].rows - 1) * L.m[1].row_bytes;
    };
    auto nb_left = [&](int u) -> int { const int unit = DUAL ? wave + (u >> 1)*NW : wave + u*NW; return Q80 ? k/32 - unit*8 : 8; };     // blocks of the row from this unit on (Q8_0: ragged last unit)
    auto unit_ptr = [&](int u) -> const char * { return DUAL ? rp[u & 1] + (size_t)(wave + (u >> 1)*NW) * BLK : rp[0] + (size_t)(wave + u*NW) * BLK; };

    mq_frag<TYPE> fa, fb;          // fb unused without PF
    int grp = blockIdx.x, mi = 0, row0 = 0;
    if (grp < total) { set_rows(grp, mi, row0); if (nu > 0) fa.load(unit_ptr(0), kq, nb_left(0)); }      // in flight across the prologue
    if (L.act.pre) {   // activation image (HBM scratch, written by the quantiser launch) -> LDS; token rows padded by 16 bytes against bank conflicts
        const int nthr = NW*WAVE, n16row = k/16;
        const i32x4 * src = (const i32x4 *) L.act.pre;
        for (int c = threadIdx.x; c < T*n16row; c += nthr) { const int t = c / n16row, o = c - t*n16row; *(i32x4 *)(lq + (size_t) t*ldq + o*16) = src[c]; }
        const float * sd = (const float *)(L.act.pre + (size_t) T*k);
        for (int c = threadIdx.x; c < T*nsb; c += nthr) ldy[c] = sd[c];
        if (!Q80) {
            const char * sr = L.act.pre + act_img_bytes(true, T, k) + (TYPE == GGML_TYPE_Q6_K ? (size_t) T*nsb*32 : 0);     // 4-byte aligned only
            for (int c = threadIdx.x; c < T*nsb*2; c += nthr) ((i32x4 *) lrec)[c] = ld16(sr + (size_t) c*16);
        }
    } else if constexpr (!Q80) {
        // in-block quantiser (K-quants, few units per wave -- mi_mmq_inline_quant): a wave per (token, super-block), straight into the LDS
        // image; saves the quantiser launch in front of this kernel while the first weight loads are already in flight
        float * sc = (float *) red;                                        // [T] RMS scales (the reduction tiles are free until the first group ends)
        if (L.act.norm) {
            for (int t = wave; t < T; t += NW) {
                const float * row = L.act.X + (size_t) t*L.act.xs;
                double s2 = 0.0;
                for (int i = lane*4; i < k; i += 256) { const float4 x = *(const float4 *)(row + i); s2 += (double)(x.x*x.x); s2 += (double)(x.y*x.y); s2 += (double)(x.z*x.z); s2 += (double)(x.w*x.w); }
                const double tot = wave_sum_d(s2);
                if (lane == 0) sc[t] = 1.0f / sqrtf((float)(tot / (double) k) + L.act.eps);
            }
            __syncthreads();
        }
        for (int u = wave; u < T*nsb; u += NW) {
            const int t = u / nsb, sb = u - t*nsb, e = sb*256 + lane*4;
            float4 v = *(const float4 *)(L.act.X + (size_t) t*L.act.xs + e);
            if (L.act.norm) {
                const float s1 = sc[t];
                v.x *= s1; v.y *= s1; v.z *= s1; v.w *= s1;
                if (L.act.norm_w) { const float4 w = *(const float4 *)(L.act.norm_w + e); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
            }
            quant_q8K_unit(v, lane, t, sb, ldq, nsb, lq, ldy, nullptr, TYPE == GGML_TYPE_Q6_K ? nullptr : lrec, TYPE == GGML_TYPE_Q6_K ? lrec : nullptr);
        }
    }
    __syncthreads();
    const mq_act A = { lq, ldq, ldy, lrec, nsb, T };

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
                if (u + 1 < nu) fb.load(unit_ptr(u + 1), kq, nb_left(u + 1));
                mq_proc<TYPE, TG>::run(fa, A, DUAL ? wave + (u >> 1)*NW : wave + u*NW, lane, acc[0]);                 // DUAL: even units = gate
                if (u + 1 >= nu) break;
                if (u + 2 < nu) fa.load(unit_ptr(u + 2), kq, nb_left(u + 2));
                mq_proc<TYPE, TG>::run(fb, A, DUAL ? wave + ((u + 1) >> 1)*NW : wave + (u + 1)*NW, lane, acc[NM - 1]);   // DUAL: odd units = up
            }
        } else {
            for (int u = 0; u < nu; u += NM) {
                mq_proc<TYPE, TG>::run(fa, A, DUAL ? wave + (u >> 1)*NW : wave + u*NW, lane, acc[0]);
                if (DUAL) { fa.load(unit_ptr(u + 1), kq, nb_left(u + 1)); mq_proc<TYPE, TG>::run(fa, A, wave + (u >> 1)*NW, lane, acc[NM - 1]); }
                if (u + NM < nu) fa.load(unit_ptr(u + NM), kq, nb_left(u + NM));
            }
        }
        // next group's first loads go out before this group's reduction / epilogue
        const int cmi = mi, crow0 = row0, gn = grp + gridDim.x;
        if (gn < total) { set_rows(gn, mi, row0); if (nu > 0) fa.load(unit_ptr(0), kq, nb_left(0)); }
        // ---- split-K reduction in a fixed order (wave 0 adds the partial tiles of waves 0..NW-1), one token group at a time
#pragma unroll
        for (int tgi = 0; tgi < TG; ++tgi) {
        if (tgi*8 >= T) break;
        __syncthreads();                                                   // wave 0 is done with the previous tiles
#pragma unroll
        for (int m = 0; m < NM; ++m) { const f32x4 v = { acc[m][tgi][0], acc[m][tgi][1], acc[m][tgi][2], acc[m][tgi][3] }; red[(m*NW + wave)*WAVE + lane] = v; }
        __syncthreads();
        if (wave == 0) {
            f32x4 v[NM];                                                    // both classes (lanes l, l^32) of all waves: every lane ends with the full sums
#pragma unroll
            for (int m = 0; m < NM; ++m) { v[m] = red[(m*NW)*WAVE + lane] + red[(m*NW)*WAVE + (lane ^ 32)]; for (int w = 1; w < NW; ++w) v[m] += red[(m*NW + w)*WAVE + lane] + red[(m*NW + w)*WAVE + (lane ^ 32)]; }
            const mmvq_mat & M = L.m[cmi];
            const int row = crow0 + (lane & 15), tg = 2*tgi + (kq & 1);
            const bool in = row < M.rows && kq < 2;
            if (DUAL) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int tok = 4*tg + r; if (in && tok < T) { const float g0 = v[0][r]; *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = (g0 / (1.0f + expf(-g0))) * v[NM - 1][r]; } }
            } else if (M.epi == EPI_F32) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int tok = 4*tg + r; if (in && tok < T) { float o = v[0][r]; if (M.res) o += M.res[(size_t) tok*M.r_tok + row]; if (M.relu) o = o > 0.f ? o : 0.f; *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = o; } }
            } else if (M.epi == EPI_F16) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const int tok = 4*tg + r; if (in && tok < T) *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(v[0][r]); }
            } else {   // RoPE (mode NORM) on the row pair held by lanes (2p, 2p+1); theta by the reference's float recurrence (ggml_rope_cache_init)
                float pr[4], th[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { pr[r] = dpp_f<DPP_XOR1>(v[0][r]); const int tok = 4*tg + r; th[r] = tok < T ? (float) L.rope.pos[tok] : 0.f; }
                const int ip = (row % L.rope.head_dim) >> 1;
                for (int j = 0; j < ip; ++j) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) th[r] *= L.rope.theta_scale;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int tok = 4*tg + r;
                    if (!(in && tok < T)) continue;
                    const float a = L.rope.freq_scale * th[r];
                    const float c = cosf(a) * L.rope.attn_factor, sn = sinf(a) * L.rope.attn_factor;
                    const float x0 = (row & 1) ? pr[r] : v[0][r], x1 = (row & 1) ? v[0][r] : pr[r];
                    const float y = (row & 1) ? x0*sn + x1*c : x0*c - x1*sn;
                    if (M.epi == EPI_ROPE_F32) *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = y;
                    else                       *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(y);
                }
            }
        }
        }   // token groups
        grp = gn;
    }
}

// ---------------------------------------------------------------- host side
static std::mutex g_mu;
static std::unordered_map<const void *, int> g_occ;         // blocks per CU, per kernel instantiation (queried with the largest LDS seen)
static std::unordered_map<const void *, size_t> g_lds;

static int blocks_per_cu(const void * fn, int threads, size_t lds) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_lds.find(fn);
    if (it == g_lds.end() || it->second < lds) {
        if (lds > 48*1024) HIP_CHECK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024));
        int nb = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds));
        if (getenv("GGML_MI355X_DEBUG_OCC")) fprintf(stderr, "[mi355x] occupancy fn=%p threads=%d lds=%zu -> %d blocks/CU\n", fn, threads, lds, nb);
        if (nb < 1) nb = 1;
        g_lds[fn] = lds; g_occ[fn] = nb;
    }
    return g_occ[fn];
}

// tuning knob (experiments only): GGML_MI355X_MMQ_CFG bit 0 = single-buffered fragments everywhere, bit 1 = never 16-wave blocks, bit 2 = one block per row group
static int mmq_cfg() { static const int v = [] { const char * e = getenv("GGML_MI355X_MMQ_CFG"); return e ? atoi(e) : 0; }(); return v; }

template <int TYPE, bool DUAL, int NW, bool PF, int TG> static void mmq_launch_one(hipStream_t st, int T, const mmvq_launch & L) {
    const size_t lds = mmq_lds_bytes(T, L.k, NW, DUAL, TYPE == GGML_TYPE_Q8_0 || TYPE == GGML_TYPE_Q4_0);
    MI_ASSERT(lds <= 160*1024 && (L.act.pre || mi_mmq_inline_quant(TYPE, T, L)));
    int total = 0;
    if (DUAL) total = (L.m[0].rows + 15) / 16;
    else for (int i = 0; i < L.n_mat; ++i) total += (L.m[i].rows + 15) / 16;
    if (total < 1) return;
    auto fn = k_mmq<TYPE, DUAL, NW, PF, TG>;
    int per_cu = blocks_per_cu((const void *) fn, NW*WAVE, lds);
    if (per_cu > 4) per_cu = 4;
    int grid = total < 256*per_cu ? total : 256*per_cu;
    if (mmq_cfg() & 4) grid = total;                                      // experiment: no persistent cap
    const int pi = mi_prof_begin(st, L, T, DUAL);
    fn<<<grid, NW*WAVE, lds, st>>>(L, T);
    mi_prof_end(st, pi);
}
template <int TYPE> static void mmq_launch_type(hipStream_t st, int T, const mmvq_launch & L) {
    const int cfg = mmq_cfg();
    int total = 0;
    for (int i = 0; i < (L.swiglu ? 1 : L.n_mat); ++i) total += (L.m[i].rows + 15) / 16;
    constexpr bool PF = TYPE != GGML_TYPE_Q6_K && TYPE != GGML_TYPE_Q8_0 && TYPE != GGML_TYPE_Q4_0;      // wide fragments: single-buffered to stay at 4 waves/SIMD
    if (T > 8) {      // several groups of 8 tokens per pass (prompt / large verification batches): weights and their unpacking are shared
        const int tg = (T + 7) / 8;
        // register budget: 3 groups (24 tokens) per pass, 2 for the dual gate|up kernel
        MI_ASSERT(tg <= (L.swiglu ? 2 : 3));
        if (L.swiglu) mmq_launch_one<TYPE, true, 8, PF, 2>(st, T, L);
        else          { if (tg == 2) mmq_launch_one<TYPE, false, 8, PF, 2>(st, T, L); else mmq_launch_one<TYPE, false, 8, PF, 3>(st, T, L); }
        return;
    }
    // few row groups (one block each): 16 waves per group, i.e. one super-block per wave at k = 4096 and <= 3 in sequence at k = 11008
    const bool wide = !(cfg & 2) && TYPE != GGML_TYPE_Q8_0 && TYPE != GGML_TYPE_Q4_0 && total <= 256 && L.k/256 >= ((cfg & 8) ? 24 : 16) && mmq_lds_bytes(T, L.k, 16, L.swiglu, false) <= 160*1024;
    if (L.swiglu) { if (wide) mmq_launch_one<TYPE, true, 16, false, 1>(st, T, L); else mmq_launch_one<TYPE, true, 8, false, 1>(st, T, L); }      // gate|up: single-buffered fragments fit two blocks per CU (21.0 -> 18.8 us)
    else          {
        if (wide) mmq_launch_one<TYPE, false, 16, PF, 1>(st, T, L);
        else if ((cfg & 16) && total > 512 && total <= 1024) mmq_launch_one<TYPE, false, 4, PF, 1>(st, T, L);       // experiment: 4-wave blocks, one round
        else if (cfg & 1) mmq_launch_one<TYPE, false, 8, false, 1>(st, T, L);
        else mmq_launch_one<TYPE, false, 8, PF, 1>(st, T, L);
    }
}

// Optional: the kernel quantises the activations itself (a wave per (token, super-block)) instead of reading an image a quantiser
// launch prepared, when there are at most GGML_MI355X_MMQ_INLINE_UPW units per wave.  OFF by default (0): measured on the 7B
// verification it costs more than the launch it saves (951 vs 1015 tokens/s at 8 units per wave: every block of wo / qkv repeats
// the quantisation).  Mirrors the block shape mmq_launch_type picks (16 waves for single-matrix launches with <= 256 row groups).
bool mi_mmq_inline_quant(int type, int T, const mmvq_launch & L) {
    static const int upw = [] { const char * e = getenv("GGML_MI355X_MMQ_INLINE_UPW"); return e ? atoi(e) : 0; }();
    if (upw <= 0 || !(type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K)) return false;
    if (T > 8 || L.act.X2 || L.k > 8192) return false;
    int total = 0;
    for (int i = 0; i < (L.swiglu ? 1 : L.n_mat); ++i) total += (L.m[i].rows + 15) / 16;
    const int nw = (!L.swiglu && total <= 256 && L.k/256 >= 16) ? 16 : 8;
    return T * (L.k/256) <= upw * nw;
}
bool mi_mmq_supported(int type, int T, int k, bool swiglu) {
    const bool b32 = type == GGML_TYPE_Q8_0 || type == GGML_TYPE_Q4_0;
    if (!(type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K || b32)) return false;
    const int tcap = type == GGML_TYPE_Q4_0 ? (swiglu ? 8 : 16) : (swiglu ? 16 : 24);      // register budget of the token-group variants
    if (T < 1 || T > tcap || k % (b32 ? 32 : 256)) return false;
    if (type == GGML_TYPE_Q4_0 && T <= 8) return false;        // measured: the dp4a kernel is faster there (21.4 vs 25.0 us at T = 6); the MFMA form pays off by halving the passes at T > 8
    return mmq_lds_bytes(T, k, 8, true, b32) <= 158*1024;
}
// most tokens one pass can take: a multiple of 8 (whole token groups) once above 8
int mi_mmq_max_tokens(int type, int k, bool swiglu) {
    for (int t = 24; t >= 8; t -= 8) if (mi_mmq_supported(type, t, k, swiglu)) return t;
    int t = 7;
    while (t > 0 && !mi_mmq_supported(type, t, k, swiglu)) --t;
    return t;
}
void mi_mmq_launch(hipStream_t st, int type, int T, const mmvq_launch & L) {
    switch (type) {
        case GGML_TYPE_Q4_K: mmq_launch_type<GGML_TYPE_Q4_K>(st, T, L); break;
        case GGML_TYPE_Q5_K: mmq_launch_type<GGML_TYPE_Q5_K>(st, T, L); break;
        case GGML_TYPE_Q6_K: mmq_launch_type<GGML_TYPE_Q6_K>(st, T, L); break;
        case GGML_TYPE_Q8_0: mmq_launch_type<GGML_TYPE_Q8_0>(st, T, L); break;
        case GGML_TYPE_Q4_0: mmq_launch_type<GGML_TYPE_Q4_0>(st, T, L); break;
        default: MI_ABORT("mmq: unsupported weight type %d", type);
    }
}
