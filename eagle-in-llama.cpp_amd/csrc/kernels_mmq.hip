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
#include "mmq_device.h"
#include <mutex>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>

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
        if (lds > 48*1024) mi_allow_big_lds(fn);
        int nb = 0;
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds));
        if (mi_lab_env("GGML_MI355X_DEBUG_OCC")) fprintf(stderr, "[mi355x] occupancy fn=%p threads=%d lds=%zu -> %d blocks/CU\n", fn, threads, lds, nb);
        if (nb < 1) nb = 1;
        g_lds[fn] = lds; g_occ[fn] = nb;
    }
    return g_occ[fn];
}

// tuning knob (experiments only): GGML_MI355X_MMQ_CFG bit 0 = single-buffered fragments everywhere, bit 1 = never 16-wave blocks, bit 2 = one block per row group
static int mmq_cfg() { static const int v = [] { const char * e = mi_lab_env("GGML_MI355X_MMQ_CFG"); return e ? atoi(e) : 0; }(); return v; }

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
    static const int upw = [] { const char * e = mi_lab_env("GGML_MI355X_MMQ_INLINE_UPW"); return e ? atoi(e) : 0; }();
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
