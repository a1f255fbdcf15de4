// kernels_bb.hip -- the big-batch product (prompts, tree verification of 25+ tokens) as an int8 GEMM on
// v_mfma_i32_32x32x32_i8.  Reference: ggml_cuda_op_mul_mat_cublas (R/ggml/src/ggml-cuda/ggml-cuda.cu:1164-1225: dequantise the whole
// matrix to fp16, then a BLAS GEMM) above 64 tokens on CDNA, mul_mat_q tiles of 128 rows x up to 128 columns below (mmq.cuh:2590,2765).
// Here the arithmetic stays the CPU backend's -- int8 activations (Q8_K / Q8_0 image written by mi_quant_act), exact integer dot
// products per sub-block, float scales -- so results keep the 2e-5 agreement with the oracle that the mat-vec kernels have.
//
// Round 2's k_mmt_bb (kernels_mmt.hip) re-used the mat-vec accumulator (16 rows x 8 tokens per wave, v_mfma_i32_16x16x64_i8 with
// half of every operand zero) and unpacked every weight tile once per 8 tokens: 2.6 % of the int8 matrix peak.  This kernel:
//   * K = 32 of the MFMA is exactly one sub-block of the K-quants / one block of Q8_0: tokens on M (32), weight rows on N (32), a
//     sub-block's integer sums arrive as C[token][row] with the ROW on the lane, so the sub-block scale is one register per lane and
//     the scaling is 16 v_mad per MFMA; nothing of an operand is padding (Q6_K's 16-element sub-blocks: two MFMAs per 32, each with the
//     other half of the weights zeroed);
//   * a wave unpacks 32 rows x 256 k of weights once per 32 tokens (round 2: once per 8) straight from the tiled layout (two 16-row tiles,
//     512-byte runs per instruction), activations go through a wave-private LDS tile (coalesced 16-byte pieces in, MFMA fragments out);
//     the token tile is a grid dimension: the tiles of a row block re-read its weights through L2 (blocks 8 apart = one XCD under
//     round-robin placement, speed only).  Two other arrangements were built and measured slower (profiles/r03_bb_per_shape.txt): token
//     tiles as an unrolled loop inside a wave (weights unpacked once per unit; hundreds of spilled registers) and one wave per token
//     tile with the block's waves sharing the weight lines in L1 (spills in the Q6_K / Q8_0 bodies, 1.3 - 2.7x slower);
//   * the mins / -32 offset terms are two MFMAs per unit against the split block sums of the image (rec32 / rec16), as in the mat-vec;
//   * 8 waves per block = RT row tiles x 8/RT k-slices (split-K, one LDS reduction at the end), so that 4096 x 4096 at 61 tokens still
//     makes 256 blocks; blocks of one row block sit 8 apart in the grid (one XCD under round-robin placement: the token tiles of a
//     row block share its weights in L2 -- speed only).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "mmq_device.h"
#include "tile_layout.h"

typedef int i32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ i32x16 mfma32(const i32x4 a, const i32x4 b) {
    const i32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, z, 0, 0, 0);
}
#define BB_LD 272                        // bytes per token row of the activation tile (256 + 16: conflict-free 16-byte fragment reads)
#define BB_WAVES 8

// C / D of the 32x32 MFMA: lane l holds column n = l & 31 (weight row), register r row m = (r & 3) + 8 (r >> 2) + 4 (l >> 5) (token)
__device__ __forceinline__ int bb_tok_of(int r, int h) { return (r & 3) + 8*(r >> 2) + 4*h; }

template <int TYPE> struct bb_traits { static constexpr bool Q80 = TYPE == GGML_TYPE_Q8_0; static constexpr int TILE = mq_tfrag<TYPE>::TILE; };

// the weight bytes a lane needs for one unit, requested at the TOP of an iteration -- ahead of the activation staging -- so that the HBM
// round trip of the weights and the L2 round trip of the activations overlap (one wait per unit instead of two in a row: 3.3 -> 2.x us)
template <int TYPE> struct bb_w;
template <> struct bb_w<GGML_TYPE_Q4_K> { i32x4 hdr, raw[4];
    __device__ __forceinline__ void load(const char * tile, int n15, int h) {
        hdr = *(const i32x4 *)(tile + 16*n15);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) raw[g4] = *(const i32x4 *)(tile + 256 + 1024*(g4 & 1) + 16*(n15 + 16*(2*(g4 >> 1) + h)));
    } };
template <> struct bb_w<GGML_TYPE_Q5_K> { i32x4 hdr, raw[4], qh;
    __device__ __forceinline__ void load(const char * tile, int n15, int h) {
        hdr = *(const i32x4 *)(tile + 16*n15);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) raw[g4] = *(const i32x4 *)(tile + 768 + 1024*(g4 & 1) + 16*(n15 + 16*(2*(g4 >> 1) + h)));
        qh = *(const i32x4 *)(tile + 256 + 16*(n15 + 16*h));
    } };
template <> struct bb_w<GGML_TYPE_Q6_K> { i32x4 scv; uint16_t dh;               // (the 6-bit fragments are fetched half a unit at a time inside bb_unit)
    __device__ __forceinline__ void load(const char * tile, int n15, int) { scv = *(const i32x4 *)(tile + 3072 + 16*n15); dh = *(const uint16_t *)(tile + 3328 + 2*n15); } };
template <> struct bb_w<GGML_TYPE_Q8_0> { i32x4 q[8], dv;
    __device__ __forceinline__ void load(const char * tile, int n15, int h) {
#pragma unroll
        for (int j = 0; j < 8; ++j) q[j] = *(const i32x4 *)(tile + 1024*(j >> 1) + 16*(n15 + 16*(2*(j & 1) + h)));
        dv = *(const i32x4 *)(tile + 4096 + 16*n15);
    } };

// one unit (256 k) of 32 rows x 32 tokens: facc[r] += contribution of this unit to out[token bb_tok_of(r, h)][row n].
// Every weight load is an aligned 16-byte item of a 512-byte run that the 32 lanes of a 16-row tile fetch together (tile_layout.h).
template <int TYPE>
__device__ __forceinline__ void bb_unit(const bb_w<TYPE> & W, const char * tile /*of this lane's row*/, const int n, const int h,
                                        const int8_t * abuf, const float * dyb, const char * recb, float (&facc)[16]) {
    const int n15 = n & 15;
    const int8_t * arow = abuf + n * BB_LD + 16*h;          // (the A operand's row is the TOKEN l & 31: same index as n)
    if constexpr (TYPE == GGML_TYPE_Q4_K || TYPE == GGML_TYPE_Q5_K) {
        // qs bytes [32 g4 + 16 h, +16) = sub-blocks 2 g4 (low nibbles) and 2 g4 + 1 (high)
        const i32x4 hdr = W.hdr;
        const i32x4 (&raw)[4] = W.raw;
        i32x4 qh = {0, 0, 0, 0};
        if constexpr (TYPE == GGML_TYPE_Q5_K) qh = W.qh;
        const uint32_t u0 = hdr.y, u1 = hdr.z, u2 = hdr.w;          // get_scale_min_k4 for all eight sub-blocks (ggml-quants.c:631-638)
        const uint32_t s_lo = u0 & 0x3f3f3f3fu, s_hi = (u2 & 0x0f0f0f0fu) | ((u0 >> 2) & 0x30303030u);
        const uint32_t m_lo = u1 & 0x3f3f3f3fu, m_hi = ((u2 >> 4) & 0x0f0f0f0fu) | ((u1 >> 2) & 0x30303030u);
        const float dw = h2f((uint16_t)(hdr.x & 0xffff)), mw = h2f((uint16_t)((uint32_t) hdr.x >> 16));
        int isum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) isum[r] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            i32x4 b = (j & 1) ? ((raw[j >> 1] >> 4) & 0x0F0F0F0F) : (raw[j >> 1] & 0x0F0F0F0F);
            if constexpr (TYPE == GGML_TYPE_Q5_K) b |= ((qh >> j) & 0x01010101) << 4;
            const i32x4 a = *(const i32x4 *)(arow + 32*j);
            const i32x16 c = mfma32(a, b);
            const int sc = byte_of(j < 4 ? s_lo : s_hi, j & 3);
#pragma unroll
            for (int r = 0; r < 16; ++r) isum[r] += __mul24(sc, c[r]);
        }
        // mins: sum_j m_j * bsum32_j with the sums split as 128 h + l (rec32: [l0..l7, 0 x 8, h0..h7, 0 x 8]), k-slots 0..7 of the lanes h = 0
        i32x4 bm = {0, 0, 0, 0};
        if (h == 0) { bm.x = (int) m_lo; bm.y = (int) m_hi; }
        const i32x4 al = h == 0 ? *(const i32x4 *)(recb + n*32) : (i32x4)(0), ah = h == 0 ? *(const i32x4 *)(recb + n*32 + 16) : (i32x4)(0);
        const i32x16 cl = mfma32(al, bm), ch = mfma32(ah, bm);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 dy = *(const f32x4 *)(dyb + 8*q + 4*h);          // tokens 8q + 4h + 0..3 = registers 4q .. 4q + 3
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int r = 4*q + e; facc[r] += (dw*dy[e])*(float) isum[r] - (mw*dy[e])*(float)(cl[r] + 128*ch[r]); }
        }
    } else if constexpr (TYPE == GGML_TYPE_Q6_K) {
        // element 128 nn + 32 q + l: ql[64 nn + 32 (q & 1) + l] nibble q >> 1, qh[32 nn + l] bits 2q, 2q + 1; 16-element sub-block 8 nn + 2 q + (l >> 4).
        // A chunk of 32 k = (nn, q); the lane half h holds l = 16 h .. 16 h + 15 = sub-block 8 nn + 2 q + h: one MFMA per sub-block, the other half zeroed
        const i32x4 scv = W.scv;
        const float dw = h2f(W.dh);
        int isum[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) isum[r] = 0;
#pragma unroll 1
        for (int nn = 0; nn < 2; ++nn) {              // (not unrolled: one half of the unit's fragments in registers at a time)
            i32x4 ql[2];
#pragma unroll
            for (int q1 = 0; q1 < 2; ++q1) ql[q1] = *(const i32x4 *)(tile + 1024*nn + 16*(n15 + 16*(2*q1 + h)));
            const i32x4 qh = *(const i32x4 *)(tile + 2048 + 16*(n15 + 16*(2*nn + h)));
            const uint32_t s01 = (uint32_t)(nn ? scv.z : scv.x), s23 = (uint32_t)(nn ? scv.w : scv.y);     // scales 8 nn + 0..3 | + 4..7
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const i32x4 lo = (q >> 1) ? ((ql[q & 1] >> 4) & 0x0F0F0F0F) : (ql[q & 1] & 0x0F0F0F0F);
                const i32x4 b = lo | (((qh >> (2*q)) & 0x03030303) << 4);
                const i32x4 a = *(const i32x4 *)(arow + 128*nn + 32*q);
                const uint32_t sw = ((q >> 1) ? s23 : s01) >> (16*(q & 1));                  // scales 8 nn + 2 q (+ 1): bytes 0 / 1
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const i32x16 c = mfma32(a, h == p ? b : (i32x4)(0));
                    const int sc = sbyte_of(sw, p);
#pragma unroll
                    for (int r = 0; r < 16; ++r) isum[r] += __mul24(sc, c[r]);
                }
                __builtin_amdgcn_sched_barrier(0);                                  // one chunk at a time: bounds the live MFMA results (16 registers each)
            }
        }
        // -32 offset: 32 * sum_s scale_s * bsum16_s (rec16: [l0..l15, h0..h15]), k-slots 0..15 of the lanes h = 0
        const i32x4 bm = h == 0 ? scv : (i32x4)(0);
        const i32x4 al = h == 0 ? *(const i32x4 *)(recb + n*32) : (i32x4)(0), ah = h == 0 ? *(const i32x4 *)(recb + n*32 + 16) : (i32x4)(0);
        const i32x16 cl = mfma32(al, bm), ch = mfma32(ah, bm);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 dy = *(const f32x4 *)(dyb + 8*q + 4*h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { const int r = 4*q + e; facc[r] += (dw*dy[e])*(float)(isum[r] - 32*(cl[r] + 128*ch[r])); }
        }
    } else {
        static_assert(TYPE == GGML_TYPE_Q8_0, "bb_unit: type");
        // a unit = 8 blocks of 32 (f16 d + 32 int8): ggml_vec_dot_q8_0_q8_0, sumf += sumi * (d_x * d_y) per block.  dyb: [block j][token]
        const i32x4 (&q)[8] = W.q;
        const i32x4 dv = W.dv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const i32x4 a = *(const i32x4 *)(arow + 32*j);
            const i32x16 c = mfma32(a, q[j]);
            const float dw = h2f((uint16_t)(((uint32_t) dv[j >> 1] >> (16*(j & 1))) & 0xffff));
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const f32x4 dy = *(const f32x4 *)(dyb + 32*j + 8*qd + 4*h);
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int r = 4*qd + e; facc[r] += (float) c[r] * (dw*dy[e]); }
            }
        }
    }
}

// grid: the token tiles of one row block (32 RT rows) 8 apart; block = 8 waves: wave = rt + RT ks, row tile rt, k slice ks (units ks, ks + KS, ..)
template <int TYPE, int RT>
__global__ void __launch_bounds__(BB_WAVES*WAVE) k_bb(const mmvq_launch L, const int T, const int n_rb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool Q80 = bb_traits<TYPE>::Q80;
    constexpr int TILE = bb_traits<TYPE>::TILE, KS = BB_WAVES / RT;
    constexpr int DYN = Q80 ? 8*32 : 32;                                 // floats of activation scales per (unit, token tile)
    constexpr int WBUF = 32*BB_LD + DYN*4 + (Q80 ? 0 : 32*32);           // a wave's staging: activations | scales | block-sum records
    const int k = L.k, nun = k/256, nsb_img = Q80 ? k/32 : nun;
    const int tid = threadIdx.x, lane = tid % WAVE, wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int b = blockIdx.x, nq = (T + 31) / 32;
    const int rb = (b & 7) + 8 * ((b >> 3) / nq), tq = (b >> 3) % nq;
    if (rb >= n_rb) return;
    const int rt = wave % RT, ks = wave / RT;
    const mmvq_mat & M = L.m[0];
    const int row0 = rb*32*RT + rt*32, t0 = tq*32;
    const int n = lane & 31, h = lane >> 5;
    // rows beyond the matrix (a 16-row tail): read a valid tile, discard the result
    const int g0 = row0 < M.rows ? row0 >> 4 : 0, g1 = row0 + 16 < M.rows ? (row0 >> 4) + 1 : g0;
    const char * tp0 = M.W + (size_t) g0 * 16 * M.row_bytes, * tp1 = M.W + (size_t) g1 * 16 * M.row_bytes;
    // image (mi_quant_act): [T][k] int8 | [T][nsb] f32 d | ([T][k/16] i16 bsums | rec32 [T][nsb][32] | rec16 [T][nsb][32])
    const char * img = L.act.pre;
    const float * img_d = (const float *)(img + (size_t) T*k);
    const char * img_rec = Q80 ? nullptr : img + act_img_bytes(true, T, k) + (TYPE == GGML_TYPE_Q6_K ? (size_t) T*nsb_img*32 : 0);
    char * wb = smem + (size_t) wave * WBUF;
    int8_t * abuf = (int8_t *) wb; float * dyb = (float *)(wb + 32*BB_LD); char * recb = wb + 32*BB_LD + DYN*4;
    float facc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) facc[r] = 0.f;
    const char * tbase = (n & 16) ? tp1 : tp0;
    bb_w<TYPE> Wn; if (ks < nun) Wn.load(tbase + (size_t) ks * TILE, n & 15, h);
    for (int u = ks; u < nun; u += KS) {
        // ---- stage the unit's activations: 32 tokens x 256 bytes as 16-byte pieces (lane -> token (l >> 4) + 4 i, piece l & 15), scales, records
        const char * tile = tbase + (size_t) u * TILE;
        const bb_w<TYPE> W = Wn;                                         // requested one iteration ago
        i32x4 av[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int tok = min(t0 + (lane >> 4) + 4*i, T - 1); av[i] = *(const i32x4 *)(img + (size_t) tok*k + u*256 + 16*(lane & 15)); }      // (token slots past the batch read the last token: their results are never stored, and a clamp is no branch)
        float dyv[Q80 ? 4 : 1]; i32x4 rv = {0, 0, 0, 0};
        if constexpr (Q80) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { const int c = lane + 64*i, tok = min(t0 + (c & 31), T - 1), j = c >> 5; dyv[i] = img_d[(size_t) tok*nsb_img + u*8 + j]; }
        } else {
            { const int tok = min(t0 + (lane & 31), T - 1); dyv[0] = img_d[(size_t) tok*nsb_img + u]; }
            { const int tok = min(t0 + (lane >> 1), T - 1); rv = ld16(img_rec + ((size_t) tok*nsb_img + u)*32 + 16*(lane & 1)); }
        }
        if (u + KS < nun) Wn.load(tbase + (size_t)(u + KS) * TILE, n & 15, h);      // the NEXT unit's weights go out behind this unit's activations (vmcnt retires in order: waiting for the activations leaves these in flight)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the previous unit's fragment reads are done (wave-private buffer: no barrier)
#pragma unroll
        for (int i = 0; i < 8; ++i) *(i32x4 *)(abuf + ((lane >> 4) + 4*i)*BB_LD + 16*(lane & 15)) = av[i];
        if constexpr (Q80) {
#pragma unroll
            for (int i = 0; i < 4; ++i) dyb[lane + 64*i] = dyv[i];                       // [block j][token]
        } else {
            if (lane < 32) dyb[lane] = dyv[0];
            *(i32x4 *)(recb + (lane >> 1)*32 + 16*(lane & 1)) = rv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        bb_unit<TYPE>(W, tile, n, h, abuf, dyb, recb, facc);
    }
    // ---- split-K: the KS waves of a row tile meet in LDS (the staging space), every wave finishes 16 / KS of the token registers
    __syncthreads();
    float * red = (float *) smem;                                        // [wave][r][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave*16 + r)*64 + lane] = facc[r];
    __syncthreads();
    constexpr int RPW = 16 / KS;                                         // registers per wave
    const int row = row0 + n;
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
        const int r = ks*RPW + i;
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s) v += red[((rt + RT*s)*16 + r)*64 + lane];
        const int tok = t0 + bb_tok_of(r, h);
        const bool act = tok < T && row < M.rows;
        if (M.epi == EPI_F32) {
            if (act) {
                if (M.res) v += M.res[(size_t) tok*M.r_tok + row];
                if (M.relu) v = v > 0.f ? v : 0.f;
                *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = v;
            }
        } else if (M.epi == EPI_F16) {                                   // wv -> the transposed f16 V cache (the CPY node folded in)
            if (act) *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(v);
        } else {
            // RoPE (mode NORM) on the row pair (2p, 2p + 1) = lanes l, l ^ 1, {cos, sin} from the table of this forward pass (graph.cpp) --
            // the expressions of the tiled mat-vec's epilogue (kernels_mmt.hip) and of k_rope
            const float pr = __shfl_xor(v, 1);
            if (act) {
                const int ip = (row % L.rope.head_dim) >> 1;
                const float2 cs = *(const float2 *)(L.rope.tab + ((size_t) tok * (L.rope.head_dim >> 1) + ip) * 2);
                const float c = cs.x, sn = cs.y;
                const float x0 = (row & 1) ? pr : v, x1 = (row & 1) ? v : pr;
                const float y = (row & 1) ? x0*sn + x1*c : x0*c - x1*sn;
                if (M.epi == EPI_ROPE_F32) *(float *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = y;
                else                       *(__half *)(M.out + (size_t) row*M.o_row + (size_t) tok*M.o_tok) = __float2half_rn(y);
            }
        }
    }
}

template <int TYPE> static size_t bb_lds() {
    constexpr bool Q80 = bb_traits<TYPE>::Q80;
    const size_t wbuf = 32*BB_LD + (Q80 ? 8*32 : 32)*4 + (Q80 ? 0 : 32*32);
    const size_t stage = BB_WAVES * wbuf, red = (size_t) BB_WAVES * 16 * 64 * 4;
    return stage > red ? stage : red;
}
template <int TYPE> static void bb_launch(hipStream_t st, int T, const mmvq_launch & L) {
    MI_ASSERT(L.act.pre && L.n_mat == 1 && L.m[0].rows % 16 == 0 && L.k % 256 == 0 && !L.m[0].ids);
    MI_ASSERT(L.m[0].epi == EPI_F32 || L.m[0].epi == EPI_F16 || (L.rope.tab && L.rope.head_dim > 0 && L.rope.head_dim % 2 == 0 && L.m[0].rows % 2 == 0));
    const int nq = (T + 31) / 32, rows = L.m[0].rows;
    static const int cus = [] { int dev = 0; hipDeviceProp_t p; if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) return 256; return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256; }();
    // two row tiles per block (4-way split-K) when that still fills the chip, else one (8-way)
    const bool two = (size_t)((rows + 63) / 64) * nq >= (size_t) cus;
    const int n_rb = two ? (rows + 63) / 64 : (rows + 31) / 32;
    const int grid = ((n_rb + 7) / 8) * 8 * nq;
    const size_t lds = bb_lds<TYPE>();
    const int pi = mi_prof_begin(st, L, T, false);
    if (two) { auto fn = k_bb<TYPE, 2>; mi_allow_big_lds((const void *) fn); fn<<<grid, BB_WAVES*WAVE, lds, st>>>(L, T, n_rb); }
    else     { auto fn = k_bb<TYPE, 1>; mi_allow_big_lds((const void *) fn); fn<<<grid, BB_WAVES*WAVE, lds, st>>>(L, T, n_rb); }
    mi_prof_end(st, pi);
}
bool mi_bb_supported(int type) {
    static const bool old = mi_lab_env("GGML_MI355X_BB_OLD") != nullptr;          // A/B: round 2's 16 x 8 accumulator kernel for every type
    return !old && (type == GGML_TYPE_Q4_K || type == GGML_TYPE_Q5_K || type == GGML_TYPE_Q6_K || type == GGML_TYPE_Q8_0);
}
void mi_bb_run(hipStream_t st, int type, int T, const mmvq_launch & L) {
    switch (type) {
        case GGML_TYPE_Q4_K: bb_launch<GGML_TYPE_Q4_K>(st, T, L); break;
        case GGML_TYPE_Q5_K: bb_launch<GGML_TYPE_Q5_K>(st, T, L); break;
        case GGML_TYPE_Q6_K: bb_launch<GGML_TYPE_Q6_K>(st, T, L); break;
        case GGML_TYPE_Q8_0: bb_launch<GGML_TYPE_Q8_0>(st, T, L); break;
        default: MI_ABORT("mi_bb_run: unsupported weight type %d", type);
    }
}
