// lane_ops.h -- cross-lane steps of wave64 reductions without LDS traffic (shared by the mat-vec, attention and element-wise kernels)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- wave64 reductions on the DPP path (no LDS traffic, unlike ds_bpermute-based __shfl):
// quad_perm xor1 / xor2, row_half_mirror, row_mirror fold a 16-lane row; the four row sums are combined through readlane.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false)); }
template <int CTRL> __device__ __forceinline__ int   dpp_i(int v)   { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
#define DPP_XOR1 0xB1      /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E      /* quad_perm [2,3,0,1] */
#define DPP_HMIR 0x141     /* row_half_mirror: lane i <-> 7-i  (acts as xor 4 once quads are uniform) */
#define DPP_MIR  0x140     /* row_mirror:      lane i <-> 15-i (acts as xor 8 once half rows are uniform) */
__device__ __forceinline__ float rdl_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }   // readlane is an int builtin: bit-cast, never convert
// Reduction steps across rows of 16 lanes and across the halves of the wave on the VALU: v_permlane16_swap / v_permlane32_swap (gfx950)
// hand every lane its own value and its partner's (lane ^ 16, lane ^ 32) as the two results, in an order that depends on the lane -- a
// commutative op takes both.  (__shfl_xor compiles to ds_bpermute: an LDS crossbar round trip per step, in the middle of latency chains.)
template <int W> __device__ __forceinline__ void lane_pair(uint32_t v, uint32_t & a, uint32_t & b) {
    if constexpr (W == 16) { const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false); a = r[0]; b = r[1]; }
    else                   { const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false); a = r[0]; b = r[1]; }
}
template <int W> __device__ __forceinline__ float max_xw(float v)  { uint32_t a, b; lane_pair<W>(__float_as_uint(v), a, b); return fmaxf(__uint_as_float(a), __uint_as_float(b)); }
template <int W> __device__ __forceinline__ float sum_xw(float v)  { uint32_t a, b; lane_pair<W>(__float_as_uint(v), a, b); return __uint_as_float(a) + __uint_as_float(b); }
template <int W> __device__ __forceinline__ int   min_xw(int v)    { uint32_t a, b; lane_pair<W>((uint32_t) v, a, b); return min((int) a, (int) b); }
template <int W> __device__ __forceinline__ double sum_xw(double v) {
    const uint2 p = *(uint2 *) &v; uint32_t a0, b0, a1, b1;
    lane_pair<W>(p.x, a0, b0); lane_pair<W>(p.y, a1, b1);
    const uint2 x = { a0, a1 }, y = { b0, b1 };
    return *(const double *) &x + *(const double *) &y;
}
__device__ __forceinline__ float row_sum_f(float v) { v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HMIR>(v); v += dpp_f<DPP_MIR>(v); return v; }
__device__ __forceinline__ float row_max_f(float v) { v = fmaxf(v, dpp_f<DPP_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_HMIR>(v)); v = fmaxf(v, dpp_f<DPP_MIR>(v)); return v; }
__device__ __forceinline__ double row_sum_d(double v) {     // sum over the 16 lanes of a DPP row, result in every lane of the row
    int2 p = *(int2 *) &v;
#define DSTEP(C) { int2 q; q.x = dpp_i<C>(p.x); q.y = dpp_i<C>(p.y); v += *(double *) &q; p = *(int2 *) &v; }
    DSTEP(DPP_XOR1) DSTEP(DPP_XOR2) DSTEP(DPP_HMIR) DSTEP(DPP_MIR)
#undef DSTEP
    return v;
}
