// kernels_ops.hip -- the small HBM/latency-bound ops of the llama / EAGLE graphs, written for
// gfx950 (wave64).  Arithmetic follows the reference CPU backend statement by statement so that
// results agree with the oracle to the last bit wherever the CPU code is not libm-dependent:
//   RMS_NORM   R/ggml/src/ggml-cpu/ggml-cpu.c:7098-7144   (sum of squares in double)
//   ROPE       R/ggml/src/ggml-cpu/ggml-cpu.c:9351-9633   (theta by the recurrence theta *= theta_scale)
//   SOFT_MAX   R/ggml/src/ggml-cpu/ggml-cpu.c:9042-9138   (scale, + slope*mask, max, exp, double sum)
//   CONCAT     R/ggml/src/ggml-cpu/ggml-cpu.c:6195-6236
//   ADD/MUL    broadcast rule of ggml_compute_forward_add/mul (src1 index = dst index % src1 extent)
// GPU-side structure replaces R/ggml/src/ggml-cuda/{norm,rope,softmax,binbcast,unary,cpy,concat,getrows}.cu
// (32-lane warps there; 64-lane waves and 16-byte accesses here).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "kernels.h"
#include "lane_ops.h"
#include <math.h>

#define WAVE 64

struct dims4 { int64_t ne[4]; int64_t nb[4]; };
static inline dims4 mk_dims(const ggml_tensor * t) {
    dims4 d; for (int i = 0; i < 4; ++i) { d.ne[i] = t->ne[i]; d.nb[i] = (int64_t) t->nb[i]; } return d;
}
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// result in every lane; the order of the additions is a fixed tree (row steps, then rows, then halves)
__device__ __forceinline__ float  wave_sum(float v)  { v += dpp_f<DPP_XOR1>(v); v += dpp_f<DPP_XOR2>(v); v += dpp_f<DPP_HMIR>(v); v += dpp_f<DPP_MIR>(v); v = sum_xw<16>(v); return sum_xw<32>(v); }
__device__ __forceinline__ double wave_sum(double v) { v = row_sum_d(v); v = sum_xw<16>(v); return sum_xw<32>(v); }
__device__ __forceinline__ float  wave_max(float v)  { v = row_max_f(v); v = max_xw<16>(v); return max_xw<32>(v); }
// block-wide reductions for blocks of NW waves (result valid in every thread)
template <typename T, int NW> __device__ __forceinline__ T block_sum(T v, T * sh) {
    v = wave_sum(v);
    const int w = threadIdx.x / WAVE, l = threadIdx.x % WAVE;
    if (l == 0) sh[w] = v;
    __syncthreads();
    T r = sh[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r += sh[i];
    __syncthreads();
    return r;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float * sh) {
    v = wave_max(v);
    const int w = threadIdx.x / WAVE, l = threadIdx.x % WAVE;
    if (l == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) r = fmaxf(r, sh[i]);
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ RMS_NORM (+ fused MUL by a weight row)
// one 256-thread block per row; y = x * (1/sqrtf(mean + eps)) [ * w ]
__global__ void __launch_bounds__(256) k_rms_norm(const char * __restrict__ x, char * __restrict__ y, const float * __restrict__ w,
                                                   int ne00, int ne01, int ne02,
                                                   int64_t nb01, int64_t nb02, int64_t nb03, int64_t nb1, int64_t nb2, int64_t nb3, float eps, int vec) {
    __shared__ double sh[4];
    const int row = blockIdx.x;
    const int i1 = row % ne01, i2 = (row / ne01) % ne02, i3 = row / (ne01 * ne02);
    const float * xr = (const float *)(x + i1*nb01 + i2*nb02 + i3*nb03);
    float       * yr = (float *)(y + i1*nb1 + i2*nb2 + i3*nb3);
    double s = 0.0;
    if (vec) {
        for (int i = threadIdx.x*4; i < ne00; i += 1024) { const float4 v = *(const float4 *)(xr + i); s += (double)(v.x*v.x); s += (double)(v.y*v.y); s += (double)(v.z*v.z); s += (double)(v.w*v.w); }
    } else {
        for (int i = threadIdx.x; i < ne00; i += 256) { const float v = xr[i]; s += (double)(v * v); }
    }
    s = block_sum<double, 4>(s, sh);
    const float mean  = (float)(s / (double) ne00);
    const float scale = 1.0f / sqrtf(mean + eps);
    if (vec) {
        for (int i = threadIdx.x*4; i < ne00; i += 1024) {
            float4 v = *(const float4 *)(xr + i);
            v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
            if (w) { const float4 q = *(const float4 *)(w + i); v.x *= q.x; v.y *= q.y; v.z *= q.z; v.w *= q.w; }
            *(float4 *)(yr + i) = v;
        }
    } else if (w) { for (int i = threadIdx.x; i < ne00; i += 256) yr[i] = (xr[i] * scale) * w[i]; }
    else          { for (int i = threadIdx.x; i < ne00; i += 256) yr[i] =  xr[i] * scale; }
}
void mi_op_rms_norm(hipStream_t st, const ggml_tensor * dst, const ggml_tensor * mul_w, const ggml_tensor * out) {
    const ggml_tensor * s0 = dst->src[0];
    const int64_t rows = s0->ne[1]*s0->ne[2]*s0->ne[3];
    if (rows == 0 || s0->ne[0] == 0) return;
    MI_ASSERT(!mul_w || mul_w->ne[0] == s0->ne[0]);
    const bool al = !((uintptr_t) s0->data & 15) && !((uintptr_t) out->data & 15) && !(mul_w && ((uintptr_t) mul_w->data & 15));
    const int vec = al && s0->ne[0] % 4 == 0 && !(s0->nb[1] & 15) && !(s0->nb[2] & 15) && !(s0->nb[3] & 15) && !(out->nb[1] & 15) && !(out->nb[2] & 15) && !(out->nb[3] & 15);
    k_rms_norm<<<(unsigned) rows, 256, 0, st>>>((const char *) s0->data, (char *) out->data, mul_w ? (const float *) mul_w->data : nullptr,
        (int) s0->ne[0], (int) s0->ne[1], (int) s0->ne[2], s0->nb[1], s0->nb[2], s0->nb[3], out->nb[1], out->nb[2], out->nb[3], mi_op_f32(dst, 0), vec);
}

// ------------------------------------------------------------------ ADD / SUB / MUL / DIV with broadcast of src1
template <int OP> __global__ void __launch_bounds__(256) k_bin_bcast(const char * __restrict__ a, const char * __restrict__ b, char * __restrict__ d,
                                                                   dims4 da, dims4 db, dims4 dd, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t i0 = i % dd.ne[0], i1 = (i / dd.ne[0]) % dd.ne[1], i2 = (i / (dd.ne[0]*dd.ne[1])) % dd.ne[2], i3 = i / (dd.ne[0]*dd.ne[1]*dd.ne[2]);
    const float x = *(const float *)(a + i0*da.nb[0] + i1*da.nb[1] + i2*da.nb[2] + i3*da.nb[3]);
    const float y = *(const float *)(b + (i0 % db.ne[0])*db.nb[0] + (i1 % db.ne[1])*db.nb[1] + (i2 % db.ne[2])*db.nb[2] + (i3 % db.ne[3])*db.nb[3]);
    float r;
    if (OP == GGML_OP_ADD) r = x + y; else if (OP == GGML_OP_SUB) r = x - y; else if (OP == GGML_OP_MUL) r = x * y; else r = x / y;
    *(float *)(d + i0*dd.nb[0] + i1*dd.nb[1] + i2*dd.nb[2] + i3*dd.nb[3]) = r;
}
void mi_op_bin_bcast(hipStream_t st, const ggml_tensor * dst) {
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    const dim3 grid(cdiv(n, 256));
    switch (dst->op) {
        case GGML_OP_ADD: k_bin_bcast<GGML_OP_ADD><<<grid, 256, 0, st>>>((const char *) a->data, (const char *) b->data, (char *) dst->data, mk_dims(a), mk_dims(b), mk_dims(dst), n); break;
        case GGML_OP_SUB: k_bin_bcast<GGML_OP_SUB><<<grid, 256, 0, st>>>((const char *) a->data, (const char *) b->data, (char *) dst->data, mk_dims(a), mk_dims(b), mk_dims(dst), n); break;
        case GGML_OP_MUL: k_bin_bcast<GGML_OP_MUL><<<grid, 256, 0, st>>>((const char *) a->data, (const char *) b->data, (char *) dst->data, mk_dims(a), mk_dims(b), mk_dims(dst), n); break;
        case GGML_OP_DIV: k_bin_bcast<GGML_OP_DIV><<<grid, 256, 0, st>>>((const char *) a->data, (const char *) b->data, (char *) dst->data, mk_dims(a), mk_dims(b), mk_dims(dst), n); break;
        default: MI_ABORT("bin_bcast: bad op %d", dst->op);
    }
}

// ------------------------------------------------------------------ UNARY (contiguous f32)
__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }   // ggml-cpu.c:1897
template <int U> __global__ void __launch_bounds__(256) k_unary(const float * __restrict__ x, float * __restrict__ y, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    float r;
    switch (U) {
        case GGML_UNARY_OP_ABS:         r = fabsf(v); break;
        case GGML_UNARY_OP_SGN:         r = (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); break;
        case GGML_UNARY_OP_NEG:         r = -v; break;
        case GGML_UNARY_OP_STEP:        r = (v > 0.f) ? 1.f : 0.f; break;
        case GGML_UNARY_OP_TANH:        r = tanhf(v); break;
        case GGML_UNARY_OP_ELU:         r = (v > 0.f) ? v : expm1f(v); break;
        case GGML_UNARY_OP_RELU:        r = (v > 0.f) ? v : 0.f; break;
        case GGML_UNARY_OP_SIGMOID:     r = 1.0f / (1.0f + expf(-v)); break;
        case GGML_UNARY_OP_GELU:        r = 0.5f*v*(1.0f + tanhf(0.79788456080286535587989211986876f*v*(1.0f + 0.044715f*v*v))); break;
        case GGML_UNARY_OP_GELU_QUICK:  r = v*(1.0f/(1.0f + expf(-1.702f*v))); break;
        case GGML_UNARY_OP_SILU:        r = silu_f(v); break;
        case GGML_UNARY_OP_HARDSWISH:   r = v * fminf(1.0f, fmaxf(0.0f, (v + 3.0f) / 6.0f)); break;
        case GGML_UNARY_OP_HARDSIGMOID: r = fminf(1.0f, fmaxf(0.0f, (v + 3.0f) / 6.0f)); break;
        case GGML_UNARY_OP_EXP:         r = expf(v); break;
        default:                        r = v;
    }
    y[i] = r;
}
void mi_op_unary(hipStream_t st, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    const float * x = (const float *) dst->src[0]->data; float * y = (float *) dst->data;
    const dim3 grid(cdiv(n, 256));
#define U_CASE(u) case u: k_unary<u><<<grid, 256, 0, st>>>(x, y, n); break;
    switch (mi_op_i32(dst, 0)) {
        U_CASE(GGML_UNARY_OP_ABS) U_CASE(GGML_UNARY_OP_SGN) U_CASE(GGML_UNARY_OP_NEG) U_CASE(GGML_UNARY_OP_STEP)
        U_CASE(GGML_UNARY_OP_TANH) U_CASE(GGML_UNARY_OP_ELU) U_CASE(GGML_UNARY_OP_RELU) U_CASE(GGML_UNARY_OP_SIGMOID)
        U_CASE(GGML_UNARY_OP_GELU) U_CASE(GGML_UNARY_OP_GELU_QUICK) U_CASE(GGML_UNARY_OP_SILU)
        U_CASE(GGML_UNARY_OP_HARDSWISH) U_CASE(GGML_UNARY_OP_HARDSIGMOID) U_CASE(GGML_UNARY_OP_EXP)
        default: MI_ABORT("unary: unsupported op %d", mi_op_i32(dst, 0));
    }
#undef U_CASE
}

__global__ void __launch_bounds__(256) k_silu_mul(const float * __restrict__ g, const float * __restrict__ u, float * __restrict__ y, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = silu_f(g[i]) * u[i];
}
void mi_op_silu_mul(hipStream_t st, const ggml_tensor * gate, const ggml_tensor * up, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    k_silu_mul<<<cdiv(n, 256), 256, 0, st>>>((const float *) gate->data, (const float *) up->data, (float *) dst->data, n);
}

__global__ void __launch_bounds__(256) k_scale(const float * __restrict__ x, float * __restrict__ y, float s, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = x[i] * s;
}
void mi_op_scale(hipStream_t st, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    k_scale<<<cdiv(n, 256), 256, 0, st>>>((const float *) dst->src[0]->data, (float *) dst->data, mi_op_f32(dst, 0), n);
}

// ------------------------------------------------------------------ CPY / CONT / DUP (any strides; f32<->f16, same-type)
template <typename TS, typename TD> __device__ __forceinline__ TD cvt(TS v);
template <> __device__ __forceinline__ float  cvt<float, float>(float v)   { return v; }
template <> __device__ __forceinline__ __half cvt<float, __half>(float v)  { return __float2half_rn(v); }
template <> __device__ __forceinline__ float  cvt<__half, float>(__half v) { return __half2float(v); }
template <> __device__ __forceinline__ __half cvt<__half, __half>(__half v){ return v; }
template <> __device__ __forceinline__ int    cvt<int, int>(int v)         { return v; }
template <typename TS, typename TD> __global__ void __launch_bounds__(256) k_cpy(const char * __restrict__ s, char * __restrict__ d, dims4 ds, dims4 dd, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t s0 = i % ds.ne[0], s1 = (i / ds.ne[0]) % ds.ne[1], s2 = (i / (ds.ne[0]*ds.ne[1])) % ds.ne[2], s3 = i / (ds.ne[0]*ds.ne[1]*ds.ne[2]);
    const int64_t d0 = i % dd.ne[0], d1 = (i / dd.ne[0]) % dd.ne[1], d2 = (i / (dd.ne[0]*dd.ne[1])) % dd.ne[2], d3 = i / (dd.ne[0]*dd.ne[1]*dd.ne[2]);
    const TS v = *(const TS *)(s + s0*ds.nb[0] + s1*ds.nb[1] + s2*ds.nb[2] + s3*ds.nb[3]);
    *(TD *)(d + d0*dd.nb[0] + d1*dd.nb[1] + d2*dd.nb[2] + d3*dd.nb[3]) = cvt<TS, TD>(v);
}
void mi_op_cpy(hipStream_t st, const ggml_tensor * src, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(src);
    if (n == 0) return;
    MI_ASSERT(n == mi_nelements(dst));
    if (src->type == dst->type && mi_is_contiguous(src) && mi_is_contiguous(dst)) {
        HIP_CHECK(hipMemcpyAsync(dst->data, src->data, mi_nbytes(src), hipMemcpyDeviceToDevice, st));
        return;
    }
    const dim3 grid(cdiv(n, 256));
    const char * s = (const char *) src->data; char * d = (char *) dst->data;
    const dims4 ds = mk_dims(src), dd = mk_dims(dst);
    if      (src->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F32) k_cpy<float, float><<<grid, 256, 0, st>>>(s, d, ds, dd, n);
    else if (src->type == GGML_TYPE_F32 && dst->type == GGML_TYPE_F16) k_cpy<float, __half><<<grid, 256, 0, st>>>(s, d, ds, dd, n);
    else if (src->type == GGML_TYPE_F16 && dst->type == GGML_TYPE_F32) k_cpy<__half, float><<<grid, 256, 0, st>>>(s, d, ds, dd, n);
    else if (src->type == GGML_TYPE_F16 && dst->type == GGML_TYPE_F16) k_cpy<__half, __half><<<grid, 256, 0, st>>>(s, d, ds, dd, n);
    else if (src->type == GGML_TYPE_I32 && dst->type == GGML_TYPE_I32) k_cpy<int, int><<<grid, 256, 0, st>>>(s, d, ds, dd, n);
    else MI_ABORT("cpy: unsupported %d -> %d", src->type, dst->type);
}

// ------------------------------------------------------------------ CONCAT (f32, any dim)
__global__ void __launch_bounds__(256) k_concat(const char * __restrict__ a, const char * __restrict__ b, char * __restrict__ d,
                                                 dims4 da, dims4 db, dims4 dd, int dim, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int64_t ix[4] = { i % dd.ne[0], (i / dd.ne[0]) % dd.ne[1], (i / (dd.ne[0]*dd.ne[1])) % dd.ne[2], i / (dd.ne[0]*dd.ne[1]*dd.ne[2]) };
    const char * p;
    if (ix[0] < da.ne[0] && ix[1] < da.ne[1] && ix[2] < da.ne[2] && ix[3] < da.ne[3]) {
        p = a + ix[0]*da.nb[0] + ix[1]*da.nb[1] + ix[2]*da.nb[2] + ix[3]*da.nb[3];
    } else {
        int64_t o[4] = {0, 0, 0, 0}; o[dim] = da.ne[dim];
        p = b + (ix[0]-o[0])*db.nb[0] + (ix[1]-o[1])*db.nb[1] + (ix[2]-o[2])*db.nb[2] + (ix[3]-o[3])*db.nb[3];
    }
    *(float *)(d + ix[0]*dd.nb[0] + ix[1]*dd.nb[1] + ix[2]*dd.nb[2] + ix[3]*dd.nb[3]) = *(const float *) p;
}
void mi_op_concat(hipStream_t st, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    k_concat<<<cdiv(n, 256), 256, 0, st>>>((const char *) dst->src[0]->data, (const char *) dst->src[1]->data, (char *) dst->data,
        mk_dims(dst->src[0]), mk_dims(dst->src[1]), mk_dims(dst), mi_op_i32(dst, 0), n);
}

// ------------------------------------------------------------------ GET_ROWS (f32 / f16 table, i32 indices, f32 out)
template <typename TS> __global__ void __launch_bounds__(256) k_get_rows(const char * __restrict__ tab, const char * __restrict__ idx, char * __restrict__ d,
                                                                       dims4 dt, dims4 di, dims4 dd, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t i0 = i % dd.ne[0], i10 = (i / dd.ne[0]) % dd.ne[1], i11 = (i / (dd.ne[0]*dd.ne[1])) % dd.ne[2], i12 = i / (dd.ne[0]*dd.ne[1]*dd.ne[2]);
    const int32_t r = *(const int32_t *)(idx + i10*di.nb[0] + i11*di.nb[1] + i12*di.nb[2]);
    const TS v = *(const TS *)(tab + i0*dt.nb[0] + (int64_t) r*dt.nb[1] + i11*dt.nb[2] + i12*dt.nb[3]);
    *(float *)(d + i0*dd.nb[0] + i10*dd.nb[1] + i11*dd.nb[2] + i12*dd.nb[3]) = cvt<TS, float>(v);
}
// the hot shape (token embeddings of a verification batch / a chain step: a few rows of a contiguous f16 table): one block per output row,
// 16 bytes of table per thread and step -- no per-element index arithmetic
__global__ void __launch_bounds__(256) k_get_rows_f16_rows(const char * __restrict__ tab, const int32_t * __restrict__ idx, char * __restrict__ d, int64_t tab_nb1, int64_t d_nb1, int ne0) {
    const int32_t r = idx[blockIdx.x];
    const __half * src = (const __half *)(tab + (int64_t) r * tab_nb1);
    float * out = (float *)(d + (int64_t) blockIdx.x * d_nb1);
    for (int i = threadIdx.x * 8; i < ne0; i += 256 * 8) {
        const float4 raw = *(const float4 *)(src + i);
        const __half2 * h = (const __half2 *) &raw;
        const float2 a = __half22float2(h[0]), b = __half22float2(h[1]), c = __half22float2(h[2]), e = __half22float2(h[3]);
        *(float4 *)(out + i) = make_float4(a.x, a.y, b.x, b.y); *(float4 *)(out + i + 4) = make_float4(c.x, c.y, e.x, e.y);
    }
}
void mi_op_get_rows(hipStream_t st, const ggml_tensor * dst) {
    const int64_t n = mi_nelements(dst);
    if (n == 0) return;
    const ggml_tensor * t = dst->src[0], * ix = dst->src[1];
    if (t->type == GGML_TYPE_F16 && t->nb[0] == 2 && dst->nb[0] == 4 && (t->ne[0] % 8) == 0 && (t->nb[1] % 16) == 0 && (dst->nb[1] % 16) == 0 && (((uintptr_t) t->data | (uintptr_t) dst->data) & 15) == 0 &&
        ix->type == GGML_TYPE_I32 && ix->nb[0] == 4 && ix->ne[1] == 1 && ix->ne[2] == 1 && ix->ne[3] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && dst->ne[2] == 1 && dst->ne[3] == 1 && dst->ne[1] == ix->ne[0] && dst->ne[1] <= 65535) {
        k_get_rows_f16_rows<<<(unsigned) dst->ne[1], 256, 0, st>>>((const char *) t->data, (const int32_t *) ix->data, (char *) dst->data, t->nb[1], dst->nb[1], (int) t->ne[0]);
        return;
    }
    const dim3 grid(cdiv(n, 256));
    if (t->type == GGML_TYPE_F32) k_get_rows<float><<<grid, 256, 0, st>>>((const char *) t->data, (const char *) ix->data, (char *) dst->data, mk_dims(t), mk_dims(ix), mk_dims(dst), n);
    else if (t->type == GGML_TYPE_F16) k_get_rows<__half><<<grid, 256, 0, st>>>((const char *) t->data, (const char *) ix->data, (char *) dst->data, mk_dims(t), mk_dims(ix), mk_dims(dst), n);
    else MI_ABORT("get_rows: unsupported table type %d", t->type);
}

// ------------------------------------------------------------------ ARGMAX (f32 rows -> i32): index of the largest value by the rule of
// ggml_vec_argmax_f32 (R/ggml/src/ggml-cpu/ggml-cpu.c:2253: the last of equal maxima).  Greedy draft / verify steps fetch 4 bytes per token instead of a logits row.
// `tab` non-NULL: the block also writes row `argmax` of a f16 / f32 table as floats (GET_ROWS(table, ARGMAX(x)) right behind: the greedy
// draft chain's token -> embedding hand-off, R/examples/eagle: the next step's input is the embedding of the token just picked)
template <typename TS>
__global__ void __launch_bounds__(1024) k_argmax(const char * __restrict__ x, int32_t * __restrict__ dst, int64_t ne0, int64_t nb1,
                                                 const char * __restrict__ tab, int64_t tab_nb1, float * __restrict__ rows, int64_t rows_nb1, int64_t row_len) {
    const float * row = (const float *)(x + (int64_t) blockIdx.x * nb1);
    // ggml_vec_argmax_f32 (R/ggml/src/ggml-cpu/ggml-cpu.c:2253-2261): max = MAX(max, x[i]); if (max == x[i]) idx = i -- among equal maxima the
    // LAST index wins (a row of -inf gives n - 1).  A NaN resets that loop's running maximum; rows holding one take the sequential form below.
    float best = -INFINITY; int bi = -1; bool nan = false;
#define AM_TAKE(v, i) { const float v_ = (v); nan |= v_ != v_; if (v_ >= best) { best = v_; bi = (i); } }      // a thread walks its indices upwards: >= keeps the last of equals
    if ((ne0 & 3) == 0 && (((uintptr_t) row) & 15) == 0) {      // 16-byte loads, four of them in flight per lane (a 32000-entry logits row: two rounds)
        const int64_t n4 = ne0 / 4;
        for (int64_t i0 = threadIdx.x; i0 < n4; i0 += 4*1024) {
            float4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int64_t i = i0 + j*1024; x[j] = i < n4 ? ((const float4 *) row)[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (i0 + j*1024 < n4) {
                    const int e = (int)((i0 + j*1024) * 4);
                    AM_TAKE(x[j].x, e) AM_TAKE(x[j].y, e + 1) AM_TAKE(x[j].z, e + 2) AM_TAKE(x[j].w, e + 3)
                }
            }
        }
    } else
    for (int64_t i = threadIdx.x; i < ne0; i += 1024) AM_TAKE(row[i], (int) i)
#undef AM_TAKE
    __shared__ float sv[16]; __shared__ int si[16]; __shared__ int sn[16];
    // wave arg-max: (value, index) pairs through the DPP row steps, then the two lane-pair steps (no LDS round trips)
#define AM_STEP(C) { const float ov = dpp_f<C>(best); const int oi = dpp_i<C>(bi); if (ov > best || (ov == best && oi > bi)) { best = ov; bi = oi; } }
    AM_STEP(DPP_XOR1) AM_STEP(DPP_XOR2) AM_STEP(DPP_HMIR) AM_STEP(DPP_MIR)
#undef AM_STEP
#define AM_PAIR(W) { uint32_t va, vb, ia, ib; lane_pair<W>(__float_as_uint(best), va, vb); lane_pair<W>((uint32_t) bi, ia, ib); \
        const float fa = __uint_as_float(va), fb = __uint_as_float(vb); \
        if (fb > fa || (fb == fa && (int) ib > (int) ia)) { best = fb; bi = (int) ib; } else { best = fa; bi = (int) ia; } }
    AM_PAIR(16) AM_PAIR(32)
#undef AM_PAIR
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool wave_nan = __any(nan);
    if (lane == 0) { sv[wave] = best; si[wave] = bi; sn[wave] = wave_nan; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool any_nan = false;
        for (int w = 0; w < 16; ++w) any_nan |= sn[w] != 0;
        for (int w = 1; w < 16; ++w) if (sv[w] > best || (sv[w] == best && si[w] > bi)) { best = sv[w]; bi = si[w]; }
        if (any_nan) {                                      // the reference loop word for word (MAX(a, b) = a > b ? a : b)
            float mx = -INFINITY; bi = 0;
            for (int64_t i = 0; i < ne0; ++i) { const float v = row[i]; mx = mx > v ? mx : v; if (mx == v) bi = (int) i; }
        }
        if (bi < 0) bi = 0;
        dst[blockIdx.x] = bi;
        si[0] = bi;
    }
    if (tab) {
        __syncthreads();
        const TS * src = (const TS *)(tab + (int64_t) si[0] * tab_nb1);
        float * out = (float *)((char *) rows + (int64_t) blockIdx.x * rows_nb1);
        for (int64_t i = threadIdx.x; i < row_len; i += 1024) out[i] = cvt<TS, float>(src[i]);
    }
}
// ARGMAX, optionally with the GET_ROWS that consumes it (`rows` = that node: table f16 / f32 [row_len, n], indices = dst, result f32 [row_len, rows of x])
bool mi_argmax_rows_supported(const ggml_tensor * dst, const ggml_tensor * rows) {
    const ggml_tensor * tab = rows->src[0], * a = dst->src[0];
    if (rows->op != GGML_OP_GET_ROWS || rows->src[1] != dst || rows->type != GGML_TYPE_F32) return false;
    if (!(tab->type == GGML_TYPE_F16 || tab->type == GGML_TYPE_F32) || tab->ne[2] != 1 || tab->ne[3] != 1 || tab->nb[0] != (tab->type == GGML_TYPE_F16 ? 2u : 4u)) return false;
    if (a->ne[2] != 1 || a->ne[3] != 1 || tab->ne[1] < a->ne[0] || rows->ne[0] != tab->ne[0] || rows->ne[1] != a->ne[1] || rows->ne[2] != 1 || rows->ne[3] != 1 || rows->nb[0] != 4) return false;
    return true;
}
void mi_op_argmax(hipStream_t st, const ggml_tensor * dst, const ggml_tensor * rows) {
    const ggml_tensor * a = dst->src[0];
    const int64_t n = mi_nrows(a);
    if (n == 0) return;
    if (!rows) { k_argmax<float><<<(unsigned) n, 1024, 0, st>>>((const char *) a->data, (int32_t *) dst->data, a->ne[0], a->nb[1], nullptr, 0, nullptr, 0, 0); return; }
    const ggml_tensor * tab = rows->src[0];
    if (tab->type == GGML_TYPE_F16) k_argmax<__half><<<(unsigned) n, 1024, 0, st>>>((const char *) a->data, (int32_t *) dst->data, a->ne[0], a->nb[1], (const char *) tab->data, tab->nb[1], (float *) rows->data, rows->nb[1], tab->ne[0]);
    else                            k_argmax<float><<<(unsigned) n, 1024, 0, st>>>((const char *) a->data, (int32_t *) dst->data, a->ne[0], a->nb[1], (const char *) tab->data, tab->nb[1], (float *) rows->data, rows->nb[1], tab->ne[0]);
}

// ------------------------------------------------------------------ top-k of logits rows on the device (second half)
// Tree drafting takes the k best candidates of every live branch at every depth (R/common/speculative.cpp:257-272: the sampler's sorted
// cur_p; R/examples/speculative/speculative-eagle.cpp:542-625 forks on cur_p[f].p).  Round 2 brought a whole logits row (128 KB) per branch
// to the host and partial_sorted 32 000 entries there: 11.6 ms of a config-3 round.  Here one block per row selects the k largest in
// descending order -- ties: the LOWER index first (the order of the host's `better`, tree_driver.cpp) -- and 8 k bytes per row come back.
// A thread keeps its elements i = tid + 1024 j in registers; every round the block takes the maximum of the 64-bit keys
// (ordered value bits << 32 | ~index) and only the winner's owner looks for its next best.
__device__ __forceinline__ uint32_t topk_ord(float v) { const uint32_t b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); }      // monotone float -> uint
__device__ __forceinline__ float    topk_val(uint32_t o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }
struct topk_rows { int32_t r[16]; };
template <int J>
__global__ void __launch_bounds__(1024) k_topk(const char * __restrict__ x, int64_t ne0, int64_t nb1, topk_rows rows, int k, int32_t * __restrict__ ids, float * __restrict__ vals) {
    const float * row = (const float *)(x + (int64_t) rows.r[blockIdx.x] * nb1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float v[J];
#pragma unroll
    for (int j = 0; j < J; ++j) { const int64_t i = tid + 1024*(int64_t) j; v[j] = i < ne0 ? row[i] : 0.f; }
    unsigned long long alive = 0;
#pragma unroll
    for (int j = 0; j < J; ++j) if (tid + 1024*(int64_t) j < ne0) alive |= 1ull << j;
    __shared__ unsigned long long wk[2][16];
    auto local_best = [&]() -> unsigned long long {
        unsigned long long best = 0;
#pragma unroll
        for (int j = 0; j < J; ++j) {
            const unsigned long long key = ((unsigned long long) topk_ord(v[j]) << 32) | (0xffffffffu - (uint32_t)(tid + 1024*j));
            if (((alive >> j) & 1) && key > best) best = key;
        }
        return best;
    };
    unsigned long long mine = local_best();
    for (int it = 0; it < k; ++it) {
        unsigned long long m = mine;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const uint32_t lo = (uint32_t) __shfl_xor((int)(uint32_t) m, off), hi = (uint32_t) __shfl_xor((int)(uint32_t)(m >> 32), off);
            const unsigned long long o = ((unsigned long long) hi << 32) | lo;
            m = o > m ? o : m;
        }
        if (lane == 0) wk[it & 1][wave] = m;
        __syncthreads();                                        // (double-buffered: one barrier per round)
        unsigned long long w = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { const unsigned long long o = wk[it & 1][i]; w = o > w ? o : w; }
        if (w == 0) {                                           // fewer than k elements in the row
            if (tid == 0) { ids[(int64_t) blockIdx.x * k + it] = -1; vals[(int64_t) blockIdx.x * k + it] = -INFINITY; }
            continue;
        }
        const uint32_t idx = 0xffffffffu - (uint32_t) w;
        if (tid == 0) { ids[(int64_t) blockIdx.x * k + it] = (int32_t) idx; vals[(int64_t) blockIdx.x * k + it] = topk_val((uint32_t)(w >> 32)); }
        if ((int)(idx & 1023) == tid) { alive &= ~(1ull << (idx >> 10)); mine = local_best(); }
    }
}
// ids / vals: DEVICE arrays [n_rows][k]; rows: host array of n_rows <= 16 row indices of `logits` (f32, contiguous rows)
bool mi_top_k_supported(const ggml_tensor * logits, int k) {
    return logits && logits->type == GGML_TYPE_F32 && logits->nb[0] == 4 && logits->ne[0] >= 1 && logits->ne[0] <= 64*1024 && k >= 1 && k <= 64 && logits->ne[2] == 1 && logits->ne[3] == 1;
}
void mi_top_k(hipStream_t st, const ggml_tensor * logits, const int32_t * rows, int n_rows, int k, int32_t * ids, float * vals) {
    MI_ASSERT(n_rows >= 1 && n_rows <= 16 && mi_top_k_supported(logits, k));
    topk_rows r{}; for (int i = 0; i < n_rows; ++i) { MI_ASSERT(rows[i] >= 0 && rows[i] < logits->ne[1]); r.r[i] = rows[i]; }
    if (logits->ne[0] <= 32*1024) k_topk<32><<<n_rows, 1024, 0, st>>>((const char *) logits->data, logits->ne[0], logits->nb[1], r, k, ids, vals);
    else                          k_topk<64><<<n_rows, 1024, 0, st>>>((const char *) logits->data, logits->ne[0], logits->nb[1], r, k, ids, vals);
}

// ------------------------------------------------------------------ ROPE (mode NORM and NEOX, f32)
struct rope_params {
    int n_dims; int mode; float freq_scale, ext_factor, attn_factor, theta_scale; float corr0, corr1;
};
__device__ __forceinline__ float rope_ramp(float low, float high, int i0) {
    const float y = (i0 / 2 - low) / fmaxf(0.001f, high - low);
    return 1.0f - fminf(1.0f, fmaxf(0.0f, y));
}
__global__ void __launch_bounds__(256) k_rope(const char * __restrict__ x, const int32_t * __restrict__ pos, const float * __restrict__ ff,
                                               char * __restrict__ y, dims4 dx, dims4 dy, rope_params p, int64_t n_pairs_total) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs_total) return;
    const int64_t hp = dx.ne[0] / 2;                       // pairs per row (incl. the un-rotated tail)
    const int64_t ip = i % hp, i1 = (i / hp) % dx.ne[1], i2 = (i / (hp*dx.ne[1])) % dx.ne[2], i3 = i / (hp*dx.ne[1]*dx.ne[2]);
    const char * xr = x + i1*dx.nb[1] + i2*dx.nb[2] + i3*dx.nb[3];
    char       * yr = y + i1*dy.nb[1] + i2*dy.nb[2] + i3*dy.nb[3];
    const int64_t i0 = 2*ip;
    if (i0 >= p.n_dims) {                                    // pass-through beyond n_dims
        const float * xs = (const float *)(xr + i0*4); float * yd = (float *)(yr + i0*4);
        yd[0] = xs[0]; yd[1] = xs[1];
        return;
    }
    // theta_{i0/2} = pos * theta_scale^(i0/2), built by the same float recurrence as ggml_rope_cache_init
    float theta = (float) pos[i2];
    for (int64_t j = 0; j < ip; ++j) theta *= p.theta_scale;
    const float th_e = theta / (ff ? ff[ip] : 1.0f);
    const float th_i = p.freq_scale * th_e;
    float th = th_i, ms = p.attn_factor;
    if (p.ext_factor != 0.0f) {
        const float mix = rope_ramp(p.corr0, p.corr1, (int) i0) * p.ext_factor;
        th = th_i * (1.0f - mix) + th_e * mix;
        ms *= 1.0f + 0.1f * logf(1.0f / p.freq_scale);
    }
    const float c = cosf(th) * ms, s = sinf(th) * ms;
    if (p.mode & GGML_ROPE_TYPE_NEOX) {
        const float * xs = (const float *)(xr + ip*4); float * yd = (float *)(yr + ip*4);
        const float x0 = xs[0], x1 = xs[p.n_dims/2];
        yd[0] = x0*c - x1*s; yd[p.n_dims/2] = x0*s + x1*c;
    } else {
        const float * xs = (const float *)(xr + i0*4); float * yd = (float *)(yr + i0*4);
        const float x0 = xs[0], x1 = xs[1];
        yd[0] = x0*c - x1*s; yd[1] = x0*s + x1*c;
    }
}
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));   // ggml.c:3697-3701
}
void mi_op_rope(hipStream_t st, const ggml_tensor * dst) {
    const ggml_tensor * x = dst->src[0], * pos = dst->src[1], * ff = dst->src[2];
    const int64_t n = mi_nelements(dst) / 2;
    if (n == 0) return;
    rope_params p;
    p.n_dims = mi_op_i32(dst, 1); p.mode = mi_op_i32(dst, 2);
    const int n_ctx_orig = mi_op_i32(dst, 4);
    const float freq_base = mi_op_f32(dst, 5); p.freq_scale = mi_op_f32(dst, 6); p.ext_factor = mi_op_f32(dst, 7);
    p.attn_factor = mi_op_f32(dst, 8);
    const float beta_fast = mi_op_f32(dst, 9), beta_slow = mi_op_f32(dst, 10);
    p.theta_scale = powf(freq_base, -2.0f / p.n_dims);
    const float start = floorf(rope_corr_dim(p.n_dims, n_ctx_orig, beta_fast, freq_base));
    const float end   = ceilf (rope_corr_dim(p.n_dims, n_ctx_orig, beta_slow, freq_base));
    p.corr0 = fmaxf(0.0f, start); p.corr1 = fminf((float)(p.n_dims - 1), end);
    k_rope<<<cdiv(n, 256), 256, 0, st>>>((const char *) x->data, (const int32_t *) pos->data, ff ? (const float *) ff->data : nullptr,
        (char *) dst->data, mk_dims(x), mk_dims(dst), p, n);
}

// ------------------------------------------------------------------ SOFT_MAX (ext): softmax(x*scale + slope*mask) per row
template <bool MASK_F16> __global__ void __launch_bounds__(256) k_soft_max(const float * __restrict__ x, const void * __restrict__ mask, float * __restrict__ y,
                                                                         int64_t nc, int64_t ne01, int64_t ne02, float scale, float max_bias, float m0, float m1, uint32_t n_head_log2) {
    __shared__ float  shf[4];
    __shared__ double shd[4];
    const int64_t row = blockIdx.x;
    const float * xr = x + row * nc;
    float       * yr = y + row * nc;
    float slope = 1.0f;
    if (max_bias > 0.0f) {
        const uint32_t h = (uint32_t)((row / ne01) % ne02);
        slope = h < n_head_log2 ? powf(m0, (float)(h + 1)) : powf(m1, (float)(2*(h - n_head_log2) + 1));
    }
    const float  * m32 = mask ? (const float  *) mask + (row % ne01) * nc : nullptr;
    const __half * m16 = mask ? (const __half *) mask + (row % ne01) * nc : nullptr;
    float mx = -INFINITY;
    for (int64_t i = threadIdx.x; i < nc; i += 256) {
        float v = __fmul_rn(xr[i], scale);
        if (mask) v = __fadd_rn(v, __fmul_rn(slope, MASK_F16 ? __half2float(m16[i]) : m32[i]));
        yr[i] = v;
        mx = fmaxf(mx, v);
    }
    mx = block_max<4>(mx, shf);
    double sum = 0.0;
    for (int64_t i = threadIdx.x; i < nc; i += 256) {
        const float e = (yr[i] == -INFINITY) ? 0.0f : expf(yr[i] - mx);
        yr[i] = e;
        sum += (double) e;
    }
    sum = block_sum<double, 4>(sum, shd);
    const float inv = (float)(1.0 / sum);
    for (int64_t i = threadIdx.x; i < nc; i += 256) yr[i] *= inv;
}
void mi_op_soft_max(hipStream_t st, const ggml_tensor * dst) {
    const ggml_tensor * x = dst->src[0], * mask = dst->src[1];
    const int64_t nc = x->ne[0], nr = x->ne[1]*x->ne[2]*x->ne[3];
    if (nc == 0 || nr == 0) return;
    const float scale = mi_op_f32(dst, 0), max_bias = mi_op_f32(dst, 1);
    const uint32_t n_head = (uint32_t) x->ne[2];
    const uint32_t n_head_log2 = 1u << (uint32_t) floorf(log2f((float) n_head));
    const float m0 = powf(2.0f, -(max_bias) / n_head_log2), m1 = powf(2.0f, -(max_bias / 2.0f) / n_head_log2);
    if (mask && mask->type == GGML_TYPE_F16)
        k_soft_max<true ><<<(unsigned) nr, 256, 0, st>>>((const float *) x->data, mask->data, (float *) dst->data, nc, x->ne[1], x->ne[2], scale, max_bias, m0, m1, n_head_log2);
    else
        k_soft_max<false><<<(unsigned) nr, 256, 0, st>>>((const float *) x->data, mask ? mask->data : nullptr, (float *) dst->data, nc, x->ne[1], x->ne[2], scale, max_bias, m0, m1, n_head_log2);
}
