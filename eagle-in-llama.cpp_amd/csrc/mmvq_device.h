// mmvq_device.h -- device helpers shared by the quantised mat-vec kernels (kernels_mmvq.hip: dp4a path,
// kernels_mmq.hip: matrix-core path): unaligned 16-byte loads, DPP wave reductions, and the activation quantisers
// (RMS-norm scale + Q8_K / Q8_0 rules of the reference CPU backend) with the layout of the activation image.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "kernels.h"

#define WAVE 64
typedef int   i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ i32x4 ld16(const void * p) { i32x4 v; __builtin_memcpy(&v, p, 16); return v; }   // any alignment
__device__ __forceinline__ float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
__device__ __forceinline__ int dot4(int a, int b, int c) { return __builtin_amdgcn_sdot4(a, b, c, false); }
__device__ __forceinline__ int dot16(i32x4 a, i32x4 b) {
    int s = dot4(a.x, b.x, 0); s = dot4(a.y, b.y, s); s = dot4(a.z, b.z, s); return dot4(a.w, b.w, s);
}
#include "lane_ops.h"
__device__ __forceinline__ float wave_sum_f(float v) {        // result in every lane
    v = row_sum_f(v);
    return (rdl_f(v, 0) + rdl_f(v, 16)) + (rdl_f(v, 32) + rdl_f(v, 48));
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = row_max_f(v);
    return fmaxf(fmaxf(rdl_f(v, 0), rdl_f(v, 16)), fmaxf(rdl_f(v, 32), rdl_f(v, 48)));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = min(v, dpp_i<DPP_XOR1>(v)); v = min(v, dpp_i<DPP_XOR2>(v)); v = min(v, dpp_i<DPP_HMIR>(v)); v = min(v, dpp_i<DPP_MIR>(v));
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    int2 p = *(int2 *) &v;
#define DSTEP(C) { int2 q; q.x = dpp_i<C>(p.x); q.y = dpp_i<C>(p.y); v += *(double *) &q; p = *(int2 *) &v; }
    DSTEP(DPP_XOR1) DSTEP(DPP_XOR2) DSTEP(DPP_HMIR) DSTEP(DPP_MIR)
#undef DSTEP
    double r = 0.0;
#pragma unroll
    for (int l = 0; l < 64; l += 16) { int2 q; q.x = __builtin_amdgcn_readlane(p.x, l); q.y = __builtin_amdgcn_readlane(p.y, l); r += *(double *) &q; }
    return r;
}

// LDS image of the quantised activations
__host__ __device__ static inline size_t act_img_bytes(bool ktype, int T, int k) {
    return ktype ? (size_t) T*k + (size_t) T*(k/256)*4 + (size_t) T*(k/16)*2 : (size_t) T*k + (size_t) T*(k/32)*4;
}
// The HBM image written by k_quant_act carries, after that prefix, two 32-byte records per (token, super-block) for the
// matrix-core kernel (K-quants only): the block sums split as s = 128*h + l with l in [0,127] so that both parts fit int8:
//   rec32: sums over 32 elements  [l0..l7, 0 x 8, h0..h7, 0 x 8]      (Q4_K / Q5_K mins term)
//   rec16: sums over 16 elements  [l0..l15, h0..h15]                  (Q6_K -32 offset term)
__host__ __device__ static inline size_t act_img_bytes_full(bool ktype, int T, int k) {
    return act_img_bytes(ktype, T, k) + (ktype ? (size_t) T*(k/256)*64 : 0);
}
static inline size_t act_lds_bytes(bool ktype, int T, int k) {
    return ktype ? (size_t) T*k + (size_t) T*(k/256)*4 + (size_t) T*(k/16)*2 : (size_t) T*k + (size_t) T*(k/32)*4;
}

// ---------------------------------------------------------------- prologue: (rms_norm * w) -> quantise X[T][k] into LDS
// Activation source of a launch (struct act_src, kernels.h).  With `norm` the block first recomputes RMS_NORM
// (+ MUL by the norm weight) of every token row itself -- sum of squares in double exactly like
// ggml_compute_forward_rms_norm_f32 -- so the normalised fp32 tensor never goes to HBM and the separate norm launch
// disappears.
template <int T, int NW> __device__ __forceinline__ void row_scales(const act_src & a, int k, float * sc /*LDS [T]*/, double * red /*LDS [NW][T]*/) {
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    double s[T];
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = 0.0;
    for (int i = threadIdx.x*4; i < k; i += NW*WAVE*4) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const float4 v = *(const float4 *)(a.X + t*a.xs + i);
            s[t] += (double)(v.x*v.x); s[t] += (double)(v.y*v.y); s[t] += (double)(v.z*v.z); s[t] += (double)(v.w*v.w);
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) { const double r = wave_sum_d(s[t]); if (lane == 0) red[wave*T + t] = r; }
    __syncthreads();
    if (threadIdx.x < T) { double tot = 0.0; for (int w = 0; w < NW; ++w) tot += red[w*T + threadIdx.x]; const float mean = (float)(tot / (double) k); sc[threadIdx.x] = 1.0f / sqrtf(mean + a.eps); }
    __syncthreads();
}
__device__ __forceinline__ float4 fetch4(const act_src & a, const float * sc, int t, int e) {
    float4 v = (a.X2 && e >= a.ksplit) ? *(const float4 *)(a.X2 + t*a.xs2 + (e - a.ksplit)) : *(const float4 *)(a.X + t*a.xs + e);
    if (a.norm) {
        const float s = sc[t];
        v.x *= s; v.y *= s; v.z *= s; v.w *= s;
        if (a.norm_w) { const float4 w = *(const float4 *)(a.norm_w + e); v.x *= w.x; v.y *= w.y; v.z *= w.z; v.w *= w.w; }
        if (a.norm_out) *(float4 *)(a.norm_out + (size_t) t*a.norm_os + e) = v;      // quantiser launches only (kernels_mmt.hip): the folded norm is materialised
    }
    return v;
}
// Q8_K rule: the scale comes from the FIRST element of largest magnitude, iscale = -127/max,
// q = min(127, rne(iscale*x)), d = 1/iscale, bsums over groups of 16.  One wave = one 256-element super-block;
// loads for PB super-blocks are issued before the first reduction so their latencies overlap.
// one 256-element super-block held by a wave (4 consecutive elements per lane) -> image of token t, super-block sb
__device__ __forceinline__ void quant_q8K_unit(const float4 v4, int lane, int t, int sb, int k, int nsb, int8_t * q, float * d, short * bs, char * rec32, char * rec16) {
    const float xv[4] = { v4.x, v4.y, v4.z, v4.w };
    float amax = 0.0f; int first = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float ax = fabsf(xv[j]); if (ax > amax) { amax = ax; first = j; } }
    const float wmax = wave_max_f(amax);
    const int key = wave_min_i((amax == wmax) ? (lane*4 + first) : (1 << 20));     // lowest index holding the maximum
    const float cand = (first == 0) ? xv[0] : (first == 1) ? xv[1] : (first == 2) ? xv[2] : xv[3];
    const float mx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cand), (key >> 2) & 63));
    int packed = 0; int s = 0; float dd = 0.0f;
    if (wmax != 0.0f) {
        const float iscale = -127.f / mx;
#pragma unroll
        for (int j = 0; j < 4; ++j) { int qi = __float2int_rn(iscale * xv[j]); qi = min(127, qi); s += qi; packed |= (qi & 0xff) << (8*j); }
        dd = 1.0f / iscale;
    }
    *(int *)(q + (size_t) t*k + sb*256 + lane*4) = packed;
    s += dpp_i<DPP_XOR1>(s); s += dpp_i<DPP_XOR2>(s);
    if (bs && (lane & 3) == 0) bs[t*(k/16) + sb*16 + (lane >> 2)] = (short) s;
    if (lane == 0) d[t*nsb + sb] = dd;
    // split sums for the matrix-core kernel (either record may be absent)
    if (rec16) { int8_t * r16 = (int8_t *) rec16 + (size_t)(t*nsb + sb)*32; if ((lane & 3) == 0) { r16[lane >> 2] = (int8_t)(s & 127); r16[16 + (lane >> 2)] = (int8_t)(s >> 7); } }
    if (rec32) {
        int8_t * r32 = (int8_t *) rec32 + (size_t)(t*nsb + sb)*32;
        const int s2 = s + dpp_i<DPP_HMIR>(s);                          // quads are uniform: the other quad of the 8 lanes
        if ((lane & 7) == 0) { r32[lane >> 3] = (int8_t)(s2 & 127); r32[16 + (lane >> 3)] = (int8_t)(s2 >> 7); }
        if ((lane & 7) == 4) { r32[8 + (lane >> 3)] = 0; r32[24 + (lane >> 3)] = 0; }
    }
}
template <int T, int NW> __device__ __forceinline__ void quant_q8K_to_lds(const act_src & a, const float * sc, int k, int8_t * q, float * d, short * bs, int ubeg, int ustr, char * rec32 = nullptr, char * rec16 = nullptr) {
    const int lane = threadIdx.x % WAVE;
    const int nsb = k / 256, nu = T*nsb;
    constexpr int PB = 4;
    for (int u0 = ubeg; u0 < nu; u0 += ustr*PB) {
        float4 vv[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) { const int u = u0 + p*ustr; if (u < nu) { const int t = u / nsb, sb = u - t*nsb; vv[p] = fetch4(a, sc, t, sb*256 + lane*4); } }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int u = u0 + p*ustr;
            if (u >= nu) break;
            const int t = u / nsb, sb = u - t*nsb;
            quant_q8K_unit(vv[p], lane, t, sb, k, nsb, q, d, bs, rec32, rec16);
        }
    }
}
// Q8_0 rule: d = amax/127, id = 1/d, q = roundf(x*id) (half away from zero), d stored through fp16.
__device__ __forceinline__ void q80_unit(const float4 v, int8_t * qdst, float * ddst, int lane, bool write) {
    const float xv[4] = { v.x, v.y, v.z, v.w };
    float amax = fmaxf(fmaxf(fabsf(xv[0]), fabsf(xv[1])), fmaxf(fabsf(xv[2]), fabsf(xv[3])));
    amax = fmaxf(amax, dpp_f<DPP_XOR1>(amax)); amax = fmaxf(amax, dpp_f<DPP_XOR2>(amax)); amax = fmaxf(amax, dpp_f<DPP_HMIR>(amax));   // 8 lanes = one block of 32
    const float dd = amax / 127.f;
    const float id = dd ? 1.0f/dd : 0.0f;
    int packed = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int qi = (int) roundf(xv[j]*id); packed |= (qi & 0xff) << (8*j); }
    if (write) { *(int *) qdst = packed; if ((lane & 7) == 0) *ddst = __half2float(__float2half_rn(dd)); }
}
template <int T, int NW> __device__ __forceinline__ void quant_q80_to_lds(const act_src & a, const float * sc, int k, int8_t * q, float * d, int ubeg, int ustr) {
    const int lane = threadIdx.x % WAVE;
    const int nch = (k + 255) / 256, nb = k / 32, nu = T*nch;       // a chunk = 256 elements = 8 blocks of 32 (last one may be ragged)
    constexpr int PB = 4;
    for (int u0 = ubeg; u0 < nu; u0 += ustr*PB) {
        float4 vv[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int u = u0 + p*ustr; vv[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (u < nu) { const int t = u / nch, ch = u - t*nch, e = ch*256 + lane*4; if (e < k) vv[p] = fetch4(a, sc, t, e); }
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) {
            const int u = u0 + p*ustr;
            if (u >= nu) break;
            const int t = u / nch, ch = u - t*nch, e = ch*256 + lane*4;
            q80_unit(vv[p], q + t*k + e, d + t*nb + e/32, lane, e < k);
        }
    }
}
