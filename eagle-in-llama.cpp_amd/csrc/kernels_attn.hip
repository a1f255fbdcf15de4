// kernels_attn.hip -- the attention sub-graph of llm_build_kqv without flash attention
// (R/src/llama.cpp:706-828) as ONE kernel for small token batches (draft steps, tree verification):
//     kq  = MUL_MAT(K f16 [d, n_kv, H_kv], q f32 [d, T, H])
//     p   = SOFT_MAX(kq * scale + mask)            mask f32 [n_kv, pad64(T)]: the tree / causal mask with -INF
//     kqv = MUL_MAT(V f16 [n_kv, d, H_kv] (transposed cache), p)
//     out = CONT(PERMUTE(kqv))  -> [d*H, T]
// replacing 2 mat-mul launches (R/ggml/src/ggml-cuda/mmv.cu:197 or batched hipBLAS ggml-cuda.cu:1693), soft_max_f32
// (softmax.cu:23) and a cpy.  Numerics follow the CPU backend: q and p are rounded to f16 before the dot products
// with the f16 cache (vec_dot_type of F16, R/ggml/src/ggml-cpu/ggml-cpu.c:260-264), products accumulate in fp32,
// the soft-max denominator in double (ggml-cpu.c:9122-9129).
//
// One 256-thread block per (head, tile of <= 8 tokens).  K rows (256 B at d = 128) are read with 16-byte loads,
// 16 lanes per cache cell; V^T rows are contiguous in the cell index, read 16 bytes per lane; scores live in LDS.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include "kernels.h"
#include <mutex>

#define WAVE 64
constexpr int ATT_TT = 8;      // tokens per block

typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h8_to_f(const i32x4 v, float * f) {
    const __half2 * h = (const __half2 *) &v;
#pragma unroll
    for (int i = 0; i < 4; ++i) { const float2 t = __half22float2(h[i]); f[2*i] = t.x; f[2*i + 1] = t.y; }
}
__device__ __forceinline__ float rnd16(float v) { return __half2float(__float2half_rn(v)); }

template <int D> __global__ void __launch_bounds__(256) k_attn_small(const mi_attn_args a) {
    extern __shared__ __attribute__((aligned(16))) float sc[];          // [TT][n_kv]
    __shared__ double shd[4];
    const int h = blockIdx.x, t0 = blockIdx.y * ATT_TT;
    const int nt = min(ATT_TT, a.T - t0);
    const int hk = h / (a.H / a.H_kv);
    const int lane = threadIdx.x % WAVE, wave = threadIdx.x / WAVE;
    const int n_kv = a.n_kv;
    const char * kb = (const char *) a.k + (int64_t) hk * a.k_nb2;
    const char * vb = (const char *) a.v + (int64_t) hk * a.v_nb2;
    constexpr int LPC = D / 8;                 // lanes per cache cell (16 at d = 128)
    constexpr int CPW = WAVE / LPC;            // cells per wave pass
    const int sub = lane / LPC, dc = lane % LPC;

    // ---- phase 1: scores
    float qr[ATT_TT][8];
#pragma unroll
    for (int t = 0; t < ATT_TT; ++t) {
        if (t < nt) {
            const float * qp = (const float *)((const char *) a.q + (int64_t)(t0 + t) * a.q_nb1 + (int64_t) h * a.q_nb2) + dc*8;
#pragma unroll
            for (int j = 0; j < 8; ++j) qr[t][j] = rnd16(qp[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) qr[t][j] = 0.f;
        }
    }
    constexpr int KU = 4;                                             // K rows in flight per lane
    for (int i0 = wave*CPW; i0 < n_kv; i0 += 4*CPW*KU) {
        i32x4 kv[KU];
#pragma unroll
        for (int u = 0; u < KU; ++u) { const int i = i0 + u*4*CPW + sub; kv[u] = (i < n_kv) ? *(const i32x4 *)(kb + (int64_t) i * a.k_nb1 + dc*16) : (i32x4)(0); }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int i = i0 + u*4*CPW + sub;
            float kf[8]; h8_to_f(kv[u], kf);
#pragma unroll
            for (int t = 0; t < ATT_TT; ++t) {
                if (t >= nt) break;
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) s += kf[j] * qr[t][j];
#pragma unroll
                for (int o = LPC/2; o > 0; o >>= 1) s += __shfl_xor(s, o, WAVE);
                if (dc == 0 && i < n_kv) sc[t*n_kv + i] = s;
            }
        }
    }
    __syncthreads();
    // ---- phase 2: soft-max per token row (one wave per row), probabilities rounded through f16
    for (int t = wave; t < nt; t += 4) {
        float * row = sc + t*n_kv;
        const float * m32 = a.mask_f16 ? nullptr : (const float *)((const char *) a.mask + (int64_t)(t0 + t) * a.mask_nb1);
        const __half * m16 = a.mask_f16 ? (const __half *)((const char *) a.mask + (int64_t)(t0 + t) * a.mask_nb1) : nullptr;
        float mx = -INFINITY;
        for (int i = lane; i < n_kv; i += WAVE) {
            float v = __fmul_rn(row[i], a.scale);
            if (a.mask) v = __fadd_rn(v, a.mask_f16 ? __half2float(m16[i]) : m32[i]);
            row[i] = v; mx = fmaxf(mx, v);
        }
#pragma unroll
        for (int o = WAVE/2; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, WAVE));
        double sum = 0.0;
        for (int i = lane; i < n_kv; i += WAVE) { const float e = (row[i] == -INFINITY) ? 0.0f : expf(row[i] - mx); row[i] = e; sum += (double) e; }
#pragma unroll
        for (int o = WAVE/2; o > 0; o >>= 1) sum += __shfl_xor(sum, o, WAVE);
        const float inv = (float)(1.0 / sum);
        for (int i = lane; i < n_kv; i += WAVE) row[i] = rnd16(row[i] * inv);
    }
    __syncthreads();
    // ---- phase 3: out[d, t] = sum_i V[i, d] * p[t][i]; a wave takes DU output dims per pass (DU independent 16-byte
    //      V loads in flight), lanes across cells (8 per lane); blockIdx.z splits the output dims between blocks
    constexpr int DU = 4;
    const int dper = D / gridDim.z, dbeg = blockIdx.z * dper;
    for (int d0 = dbeg + wave*DU; d0 < dbeg + dper; d0 += 4*DU) {
        float acc[DU][ATT_TT];
#pragma unroll
        for (int u = 0; u < DU; ++u)
#pragma unroll
            for (int t = 0; t < ATT_TT; ++t) acc[u][t] = 0.f;
        for (int i = lane*8; i < n_kv; i += WAVE*8) {
            i32x4 vv[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) vv[u] = *(const i32x4 *)(vb + (int64_t)(d0 + u) * a.v_nb1 + (int64_t) i * 2);
#pragma unroll
            for (int t = 0; t < ATT_TT; ++t) {
                if (t >= nt) break;
                const float4 p0 = *(const float4 *)(sc + t*n_kv + i), p1 = *(const float4 *)(sc + t*n_kv + i + 4);
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    float vf[8]; h8_to_f(vv[u], vf);
                    acc[u][t] += vf[0]*p0.x + vf[1]*p0.y + vf[2]*p0.z + vf[3]*p0.w + vf[4]*p1.x + vf[5]*p1.y + vf[6]*p1.z + vf[7]*p1.w;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < DU; ++u)
#pragma unroll
            for (int t = 0; t < ATT_TT; ++t) {
                if (t >= nt) break;
                float v = acc[u][t];
#pragma unroll
                for (int o = WAVE/2; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
                if (lane == 0) *(float *)((char *) a.out + (int64_t)(d0 + u)*4 + (int64_t) h * a.o_nb1 + (int64_t)(t0 + t) * a.o_nb2) = v;
            }
    }
    (void) shd;
}

bool mi_attn_small_supported(const mi_attn_args & a) {
    if (!(a.d == 64 || a.d == 128)) return false;
    if (a.n_kv % 8 || a.n_kv <= 0 || (size_t) a.n_kv * ATT_TT * 4 > 150*1024) return false;
    if (a.H % a.H_kv) return false;
    if (((uintptr_t) a.k | (uintptr_t) a.v | (uintptr_t) a.k_nb1 | (uintptr_t) a.k_nb2 | (uintptr_t) a.v_nb1 | (uintptr_t) a.v_nb2) & 15) return false;
    if (((uintptr_t) a.q | (uintptr_t) a.q_nb1 | (uintptr_t) a.q_nb2) & 3) return false;
    return true;
}

void mi_op_attn_small(hipStream_t st, const mi_attn_args & a) {
    const size_t lds = (size_t) a.n_kv * ATT_TT * 4;
    const int tiles = (a.T + ATT_TT - 1) / ATT_TT;
    const int dsplit = (a.H * tiles >= 256) ? 1 : ((a.H * tiles >= 128) ? 2 : 4);     // more blocks when heads x tiles under-fill the chip
    const dim3 grid(a.H, tiles, dsplit);
    static std::once_flag once;
    std::call_once(once, [] {
        HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_small<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 152*1024));
        HIP_CHECK(hipFuncSetAttribute((const void *) k_attn_small<64>,  hipFuncAttributeMaxDynamicSharedMemorySize, 152*1024));
    });
    if (a.d == 128) k_attn_small<128><<<grid, 256, lds, st>>>(a);
    else            k_attn_small<64><<<grid, 256, lds, st>>>(a);
}
