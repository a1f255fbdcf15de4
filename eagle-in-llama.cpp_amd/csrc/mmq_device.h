// mmq_device.h -- device code shared by the matrix-core mat-vec kernels (kernels_mmq.hip: ggml's row-major blocks; kernels_mmt.hip:
// the tiled re-layout of tile_layout.h): weight fragments (what a lane keeps of one 16-row x 1-unit tile), the LDS view of the
// quantised activation image, and mq_proc -- the integer dot products of ggml_vec_dot_{q4_K,q5_K,q6_K}_q8_K / {q8_0,q4_0}_q8_0
// (R/ggml/src/ggml-cpu/ggml-cpu-quants.c) on v_mfma_i32_16x16x64_i8.  See kernels_mmq.hip for the lane maps.
#pragma once
#include "mmvq_device.h"

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
    __device__ __forceinline__ i32x4 get_ql(int nn) const { return fix16(ql[nn], sh); }
    __device__ __forceinline__ i32x4 get_qh() const { return fix16(qh, sh); }
    __device__ __forceinline__ i32x4 get_sc() const { return fix16(sc, sh); }
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
    __device__ __forceinline__ i32x4 get_q(int a) const { return fix16(q[a], ((shs >> a) & 1) * 16); }
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
    __device__ __forceinline__ i32x4 get_q(int a) const { return fix16(q[a], ((shs >> a) & 1) * 16); }
};


// ---- the same fragments from the TILED layout (tile_layout.h): `t` = base of the 16-row x 1-unit tile; every load is an aligned
// 16-byte piece of a 1 KiB (or 256 / 512 B) run that the 64 lanes of the wave fetch together
// NT: the mat-vec kernels stream every weight tile exactly once per launch (one CU, one read): requested non-temporally they do not
// allocate in L2 / MALL and the stream runs ~6 % faster end to end (profiles/r02_ab_second_half.txt); the big-batch GEMM, whose four
// token quarters share a tile through L2, keeps the default policy
template <bool NT> __device__ __forceinline__ i32x4 ldw(const i32x4 * p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <int TYPE> struct mq_tfrag;
template <> struct mq_tfrag<GGML_TYPE_Q4_K> {
    static constexpr int TILE = 2304;
    i32x4 hdr, qs[2];
    template <bool NT = true> __device__ __forceinline__ void load(const char * t, int lane, int) {
        hdr = ldw<NT>((const i32x4 *)(t + 16*(lane & 15))); qs[0] = ldw<NT>((const i32x4 *)(t + 256 + 16*lane)); qs[1] = ldw<NT>((const i32x4 *)(t + 1280 + 16*lane));
    }
};
template <> struct mq_tfrag<GGML_TYPE_Q5_K> {
    static constexpr int TILE = 2816;
    i32x4 hdr, qh, qs[2];
    template <bool NT = true> __device__ __forceinline__ void load(const char * t, int lane, int) {
        hdr = ldw<NT>((const i32x4 *)(t + 16*(lane & 15))); qh = ldw<NT>((const i32x4 *)(t + 256 + 16*((lane & 15) + 16*((lane >> 4) & 1))));
        qs[0] = ldw<NT>((const i32x4 *)(t + 768 + 16*lane)); qs[1] = ldw<NT>((const i32x4 *)(t + 1792 + 16*lane));
    }
};
template <> struct mq_tfrag<GGML_TYPE_Q6_K> {
    static constexpr int TILE = 3360;
    i32x4 ql[2], qh, sc; int dh;
    template <bool NT = true> __device__ __forceinline__ void load(const char * t, int lane, int) {
        ql[0] = ldw<NT>((const i32x4 *)(t + 16*lane)); ql[1] = ldw<NT>((const i32x4 *)(t + 1024 + 16*lane)); qh = ldw<NT>((const i32x4 *)(t + 2048 + 16*lane));
        sc = ldw<NT>((const i32x4 *)(t + 3072 + 16*(lane & 15))); dh = *(const uint16_t *)(t + 3328 + 2*(lane & 15));
    }
    __device__ __forceinline__ i32x4 get_ql(int nn) const { return ql[nn]; }
    __device__ __forceinline__ i32x4 get_qh() const { return qh; }
    __device__ __forceinline__ i32x4 get_sc() const { return sc; }
};
template <> struct mq_tfrag<GGML_TYPE_Q8_0> {          // k % 256 == 0: every unit has its eight blocks
    static constexpr int TILE = 4352;
    i32x4 q[4]; int dh[4];
    template <bool NT = true> __device__ __forceinline__ void load(const char * t, int lane, int) {
#pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = ldw<NT>((const i32x4 *)(t + 1024*a + 16*lane));
        const i32x4 d = ldw<NT>((const i32x4 *)(t + 4096 + 16*(lane & 15)));                 // the eight f16 block scales of this lane's row
        const int hi = (lane >> 5) & 1;                                               // block 2a + (kq>>1): halfword hi of dword a
#pragma unroll
        for (int a = 0; a < 4; ++a) dh[a] = ((uint32_t) d[a] >> (16*hi)) & 0xffff;
    }
    __device__ __forceinline__ i32x4 get_q(int a) const { return q[a]; }
};
template <> struct mq_tfrag<GGML_TYPE_Q4_0> {
    static constexpr int TILE = 2304;
    i32x4 q[4]; int dh[4];
    template <bool NT = true> __device__ __forceinline__ void load(const char * t, int lane, int) {
        const int l32 = (lane & 15) + 16*((lane >> 5) & 1);                           // (row, kq>>1): both nibble halves read the same 16 bytes
#pragma unroll
        for (int a = 0; a < 4; ++a) q[a] = ldw<NT>((const i32x4 *)(t + 512*a + 16*l32));
        const i32x4 d = ldw<NT>((const i32x4 *)(t + 2048 + 16*(lane & 15)));
        const int hi = (lane >> 5) & 1;
#pragma unroll
        for (int a = 0; a < 4; ++a) dh[a] = ((uint32_t) d[a] >> (16*hi)) & 0xffff;
    }
    __device__ __forceinline__ i32x4 get_q(int a) const { return q[a]; }
};

// LDS view of the activation image
struct mq_act { const int8_t * q; int ldq; const float * d; const char * rec; int nsb; int T; };   // nsb: scales per token (super-blocks; 32-blocks for Q8_0)

// ---- one super-block of 16 rows x T tokens.  The M dimension of the MFMA carries (token, class): M rows 0..7 are the
// tokens against the k-slots of one sub-block, M rows 8..15 the same tokens against another sub-block, the activations
// being zero in the k-slots of the other class.  C lane (n, g = lane>>4), reg r: M row 4g + r, i.e. class g>>1, token
// 4*(g&1) + r: acc[r] is this lane's share (its class) of out[row n][token 4*(g&1) + r]; classes are added in the reduction.
// TG groups of 8 tokens per pass: the B operands (unpacked quants) are built once and multiplied with TG activation operands.
template <int TYPE, int TG> struct mq_proc;       // run() is generic over the fragment class F (row-major mq_frag or tiled mq_tfrag)

template <int TYPE, int TG, class F> __device__ __forceinline__ void mq_process_q45(const F & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) {
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
template <int TG> struct mq_proc<GGML_TYPE_Q4_K, TG> { template <class F> static __device__ __forceinline__ void run(const F & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) { mq_process_q45<GGML_TYPE_Q4_K, TG, F>(f, A, sb, lane, acc); } };
template <int TG> struct mq_proc<GGML_TYPE_Q5_K, TG> { template <class F> static __device__ __forceinline__ void run(const F & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) { mq_process_q45<GGML_TYPE_Q5_K, TG, F>(f, A, sb, lane, acc); } };

template <int TG> struct mq_proc<GGML_TYPE_Q6_K, TG> { template <class F> static __device__ __forceinline__ void run(const F & f, const mq_act & A, int sb, int lane, float (&acc)[TG][4]) {
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
    const i32x4 qh_own = f.get_qh(), scv = f.get_sc();
#pragma unroll
    for (int nn = 0; nn < 2; ++nn) {
        // this lane needs qh bytes [32nn + 16lh, +16): they were loaded by lane group kq' = 2nn + lh of the same row
        const int src = 4*((lane & 15) + 16*(2*nn + lh));
        i32x4 qhn;
        qhn.x = __builtin_amdgcn_ds_bpermute(src, qh_own.x); qhn.y = __builtin_amdgcn_ds_bpermute(src, qh_own.y);
        qhn.z = __builtin_amdgcn_ds_bpermute(src, qh_own.z); qhn.w = __builtin_amdgcn_ds_bpermute(src, qh_own.w);
        const i32x4 hq = qhn >> (2*qb);                                    // bits 2q.. for q = qb (nib 0) at bit 0, for q = qb + 2 (nib 1) at bit 4
        const i32x4 qln = f.get_ql(nn);
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

template <int TG> struct mq_proc<GGML_TYPE_Q8_0, TG> { template <class F> static __device__ __forceinline__ void run(const F & f, const mq_act & A, int unit, int lane, float (&acc)[TG][4]) {
    // ggml_vec_dot_q8_0_q8_0: sumf += sumi * (d_x * d_y) per 32-element block; activations quantised with quantize_row_q8_0 (image d per block)
    const int i = lane & 15, kq = lane >> 4, g = kq;
    const int tok_a = i & 7, cls_a = i >> 3, cls = g >> 1;
    const bool cv = cls_a == (kq >> 1);
    const int nblk = A.nsb;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const i32x4 b = f.get_q(a);
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

template <int TG> struct mq_proc<GGML_TYPE_Q4_0, TG> { template <class F> static __device__ __forceinline__ void run(const F & f, const mq_act & A, int unit, int lane, float (&acc)[TG][4]) {
    // ggml_vec_dot_q4_0_q8_0: sumi = sum (q - 8) * a per 32-element block, sumf += (sumi * d_x) * d_y.  sum q*a and sum a come from two
    // MFMAs against the same activation operand (the second one with an all-ones B), so no block sums are needed in the image.
    const int i = lane & 15, kq = lane >> 4, g = kq;
    const int tok_a = i & 7, cls_a = i >> 3, cls = g >> 1;
    const bool cv = cls_a == (kq >> 1);
    const int nblk = A.nsb;
    const i32x4 ones = (i32x4)(0x01010101);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const i32x4 raw = f.get_q(a);
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
