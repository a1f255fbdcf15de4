// kernels_mmf.hip -- GGML_OP_MUL_MAT with an unquantised src0 (f16 / bf16 / f32) and f32 src1:
// the attention products K.q and V.p on the fp16 KV cache when they are not fused, and
// unquantised weights.  Replaces R/ggml/src/ggml-cuda/mmv.cu:5 (mul_mat_vec) and the batched
// hipBLAS path R/ggml/src/ggml-cuda/ggml-cuda.cu:1693 for the shapes the llama/eagle graphs emit.
//
// Parity with the CPU backend (R/ggml/src/ggml-cpu/ggml-cpu.c:7526-7718): the CPU converts src1 to
// src0's vec_dot_type first -- f16 for an f16 src0 (ggml-cpu.c:260-264), bf16 for bf16 -- and
// accumulates the products in fp32 (ggml_vec_dot_f16 :1539).  We round src1 the same way.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "kernels.h"

#define WAVE 64
constexpr int MMF_NC = 4;   // src1 columns per wave

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t) h << 16); }
// ggml_compute_fp32_to_bf16 (R/ggml/src/ggml-impl.h): round-to-nearest-even, quiet NaNs
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffff) > 0x7f800000) return (uint16_t)((u >> 16) | 64);
    return (uint16_t)((u + (0x7fff + ((u >> 16) & 1))) >> 16);
}

template <int ST> __device__ __forceinline__ float ld_w(const char * p, int64_t i);
template <> __device__ __forceinline__ float ld_w<GGML_TYPE_F32 >(const char * p, int64_t i) { return ((const float *) p)[i]; }
template <> __device__ __forceinline__ float ld_w<GGML_TYPE_F16 >(const char * p, int64_t i) { return __half2float(((const __half *) p)[i]); }
template <> __device__ __forceinline__ float ld_w<GGML_TYPE_BF16>(const char * p, int64_t i) { return bf16_to_f32(((const uint16_t *) p)[i]); }
template <int ST> __device__ __forceinline__ float rnd_x(float v);
template <> __device__ __forceinline__ float rnd_x<GGML_TYPE_F32 >(float v) { return v; }
template <> __device__ __forceinline__ float rnd_x<GGML_TYPE_F16 >(float v) { return __half2float(__float2half_rn(v)); }
template <> __device__ __forceinline__ float rnd_x<GGML_TYPE_BF16>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <int ST> __global__ void __launch_bounds__(256) k_mmf(const char * __restrict__ A, const char * __restrict__ B, char * __restrict__ D,
        int64_t k, int64_t ne01, int64_t ne11, int64_t ne12, int64_t r2, int64_t r3,
        int64_t nb01, int64_t nb02, int64_t nb03, int64_t nb11, int64_t nb12, int64_t nb13, int64_t nb1, int64_t nb2, int64_t nb3) {
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int64_t row = (int64_t) blockIdx.x * 4 + wave;
    if (row >= ne01) return;
    const int64_t c0 = (int64_t) blockIdx.y * MMF_NC;
    const int64_t i2 = blockIdx.z % ne12, i3 = blockIdx.z / ne12;
    const char * a = A + row*nb01 + (i2/r2)*nb02 + (i3/r3)*nb03;
    const char * b = B + i2*nb12 + i3*nb13;
    float acc[MMF_NC];
#pragma unroll
    for (int c = 0; c < MMF_NC; ++c) acc[c] = 0.f;
    for (int64_t i = lane; i < k; i += WAVE) {
        const float w = ld_w<ST>(a, i);
#pragma unroll
        for (int c = 0; c < MMF_NC; ++c) {
            const int64_t col = c0 + c < ne11 ? c0 + c : ne11 - 1;
            acc[c] += w * rnd_x<ST>(((const float *)(b + col*nb11))[i]);
        }
    }
#pragma unroll
    for (int c = 0; c < MMF_NC; ++c) {
        float v = acc[c];
#pragma unroll
        for (int o = WAVE/2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
        if (lane == 0 && c0 + c < ne11) *(float *)(D + row*4 + (c0 + c)*nb1 + i2*nb2 + i3*nb3) = v;
    }
}

void mi_op_mul_mat_f(hipStream_t st, const ggml_tensor * dst) {
    const ggml_tensor * a = dst->src[0], * b = dst->src[1];
    const int64_t k = a->ne[0];
    if (mi_nelements(dst) == 0) return;
    const dim3 grid((unsigned)((a->ne[1] + 3) / 4), (unsigned)((b->ne[1] + MMF_NC - 1) / MMF_NC), (unsigned)(b->ne[2]*b->ne[3]));
    const int64_t r2 = b->ne[2] / a->ne[2], r3 = b->ne[3] / a->ne[3];
#define MMF_ARGS (const char *) a->data, (const char *) b->data, (char *) dst->data, k, a->ne[1], b->ne[1], b->ne[2], r2, r3, \
        (int64_t) a->nb[1], (int64_t) a->nb[2], (int64_t) a->nb[3], (int64_t) b->nb[1], (int64_t) b->nb[2], (int64_t) b->nb[3], \
        (int64_t) dst->nb[1], (int64_t) dst->nb[2], (int64_t) dst->nb[3]
    switch (a->type) {
        case GGML_TYPE_F32:  k_mmf<GGML_TYPE_F32 ><<<grid, 256, 0, st>>>(MMF_ARGS); break;
        case GGML_TYPE_F16:  k_mmf<GGML_TYPE_F16 ><<<grid, 256, 0, st>>>(MMF_ARGS); break;
        case GGML_TYPE_BF16: k_mmf<GGML_TYPE_BF16><<<grid, 256, 0, st>>>(MMF_ARGS); break;
        default: MI_ABORT("mul_mat_f: unsupported src0 type %d", a->type);
    }
#undef MMF_ARGS
}
