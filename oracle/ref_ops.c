/*
 * oracle/ref_ops.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ref_quants.c for the rules and the
 * parity status: PINNED against oracle/_ref/libggml-ref-scalar.so by tests/test_oracle.py).
 *
 * Whole-op restatements on plain contiguous arrays, ggml dimension order (ne0 fastest).
 * R = /root/reference/llama.cpp, file ggml/src/ggml-cpu/ggml-cpu.c unless noted.
 */
#include "oracle.h"
#define _GNU_SOURCE
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* GGML_OP_MUL_MAT, quantised src0: ggml_compute_forward_mul_mat :7526-7718.  src1 rows are first
 * converted to the weight type's vec_dot_type (type table :254-400: Q8_0 for Q4_0/Q8_0, Q8_K for the
 * K-quants), then every dst element is ONE vec_dot over the full k. */
void orc_mul_mat_q(int wtype, const void * w, const float * x, float * dst, int64_t k, int64_t rows, int64_t T) {
    const size_t wrow = (size_t)(k / orc_type_block(wtype)) * orc_type_size(wtype);
    const int ktype = (wtype == ORC_Q4_K || wtype == ORC_Q5_K || wtype == ORC_Q6_K);
    void * yq = ktype ? malloc((size_t)(k/256) * sizeof(orc_block_q8_K)) : malloc((size_t)(k/32) * sizeof(orc_block_q8_0));
    for (int64_t t = 0; t < T; ++t) {
        if (ktype) orc_quantize_row_q8_K(x + t*k, (orc_block_q8_K *) yq, k);
        else       orc_quantize_row_q8_0(x + t*k, (orc_block_q8_0 *) yq, k);
        #pragma omp parallel for schedule(static)
        for (int64_t r = 0; r < rows; ++r) {
            const char * wr = (const char *) w + (size_t) r * wrow;
            float v = 0;
            switch (wtype) {
                case ORC_Q4_0: v = orc_vec_dot_q4_0_q8_0(k, (const orc_block_q4_0 *) wr, (const orc_block_q8_0 *) yq); break;
                case ORC_Q8_0: v = orc_vec_dot_q8_0_q8_0(k, (const orc_block_q8_0 *) wr, (const orc_block_q8_0 *) yq); break;
                case ORC_Q4_K: v = orc_vec_dot_q4_K_q8_K(k, (const orc_block_q4_K *) wr, (const orc_block_q8_K *) yq); break;
                case ORC_Q5_K: v = orc_vec_dot_q5_K_q8_K(k, (const orc_block_q5_K *) wr, (const orc_block_q8_K *) yq); break;
                case ORC_Q6_K: v = orc_vec_dot_q6_K_q8_K(k, (const orc_block_q6_K *) wr, (const orc_block_q8_K *) yq); break;
                default: break;
            }
            dst[t*rows + r] = v;
        }
    }
    free(yq);
}
/* f16 src0: src1 is rounded to f16 (vec_dot_type of F16 is F16, :260-264) and ggml_vec_dot_f16 (:1539,
 * scalar branch) sums float products in a double accumulator. */
void orc_mul_mat_f16(const uint16_t * a, int64_t a_row_stride, const float * x, float * dst, int64_t k, int64_t rows, int64_t T) {
    uint16_t * xh = (uint16_t *) malloc((size_t) k * 2);
    for (int64_t t = 0; t < T; ++t) {
        for (int64_t i = 0; i < k; ++i) xh[i] = orc_fp32_to_fp16(x[t*k + i]);
        for (int64_t r = 0; r < rows; ++r) {
            double s = 0.0;
            for (int64_t i = 0; i < k; ++i) s += (double)(orc_fp16_to_fp32(a[r*a_row_stride + i]) * orc_fp16_to_fp32(xh[i]));
            dst[t*rows + r] = (float) s;
        }
    }
    free(xh);
}
/* f32 src0: ggml_vec_dot_f32 scalar branch (:1480): double accumulator */
void orc_mul_mat_f32(const float * a, const float * x, float * dst, int64_t k, int64_t rows, int64_t T) {
    for (int64_t t = 0; t < T; ++t) for (int64_t r = 0; r < rows; ++r) {
        double s = 0.0;
        for (int64_t i = 0; i < k; ++i) s += (double)(a[r*k + i] * x[t*k + i]);
        dst[t*rows + r] = (float) s;
    }
}
/* GGML_OP_RMS_NORM: ggml_compute_forward_rms_norm_f32 :7098-7144 */
void orc_rms_norm(const float * x, float * y, int64_t ne0, int64_t nrows, float eps) {
    for (int64_t r = 0; r < nrows; ++r) {
        const float * xr = x + r*ne0; float * yr = y + r*ne0;
        double sum = 0.0;
        for (int64_t i = 0; i < ne0; ++i) sum += (double)(xr[i]*xr[i]);
        const float mean = (float)(sum/ne0);
        const float scale = 1.0f/sqrtf(mean + eps);
        for (int64_t i = 0; i < ne0; ++i) yr[i] = xr[i]*scale;
    }
}
/* GGML_OP_ROPE: ggml_compute_forward_rope_f32 :9449-9633, cache init :9375-9390, rope_yarn :9351-9372,
 * corr dims R/ggml/src/ggml.c:3697-3711.  x is [ne0, ne1 heads, ne2 tokens]. */
static float yarn_ramp(float low, float high, int i0) { const float y = (i0/2 - low)/fmaxf(0.001f, high - low); return 1 - fminf(1, fmaxf(0, y)); }
static float corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) { return n_dims*logf(n_ctx_orig/(n_rot*2*(float) M_PI))/(2*logf(base)); }
void orc_rope(const float * x, const int32_t * pos, float * y, int64_t ne0, int64_t ne1, int64_t ne2, int n_dims, int mode,
              float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, int n_ctx_orig) {
    const float theta_scale = powf(freq_base, -2.0f/n_dims);
    float corr[2];
    corr[0] = fmaxf(0, floorf(corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base)));
    corr[1] = fminf((float)(n_dims - 1), ceilf(corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base)));
    float * cache = (float *) malloc((size_t) ne0 * 4);
    for (int64_t i2 = 0; i2 < ne2; ++i2) {
        float theta = (float) pos[i2];
        for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
            const float te = theta, ti = freq_scale*te;
            float th = ti, ms = attn_factor;
            if (ext_factor != 0.0f) { const float mix = yarn_ramp(corr[0], corr[1], (int) i0)*ext_factor; th = ti*(1 - mix) + te*mix; ms *= 1.0f + 0.1f*logf(1.0f/freq_scale); }
            cache[i0] = cosf(th)*ms; cache[i0 + 1] = sinf(th)*ms;
            theta *= theta_scale;
        }
        for (int64_t i1 = 0; i1 < ne1; ++i1) {
            const float * s = x + (i2*ne1 + i1)*ne0; float * d = y + (i2*ne1 + i1)*ne0;
            if (mode & 2) {           /* NEOX */
                for (int64_t i0 = 0; i0 < n_dims; i0 += 2) { const int64_t ic = i0/2; const float c = cache[i0], sn = cache[i0+1], x0 = s[ic], x1 = s[ic + n_dims/2]; d[ic] = x0*c - x1*sn; d[ic + n_dims/2] = x0*sn + x1*c; }
            } else {
                for (int64_t i0 = 0; i0 < n_dims; i0 += 2) { const float c = cache[i0], sn = cache[i0+1], x0 = s[i0], x1 = s[i0+1]; d[i0] = x0*c - x1*sn; d[i0+1] = x0*sn + x1*c; }
            }
            for (int64_t i0 = n_dims; i0 < ne0; ++i0) d[i0] = s[i0];
        }
    }
    free(cache);
}
/* GGML_OP_SOFT_MAX (ext, max_bias = 0): ggml_compute_forward_soft_max_f32 :9042-9138 with
 * ggml_vec_soft_max_f32 scalar branch (expf, double sum).  x is [nc, ne01, ne02]; mask [nc, >= ne01]. */
void orc_soft_max(const float * x, const float * mask, float * y, int64_t nc, int64_t ne01, int64_t ne02, float scale) {
    for (int64_t r = 0; r < ne01*ne02; ++r) {
        const float * xr = x + r*nc; float * yr = y + r*nc;
        const float * mr = mask ? mask + (r % ne01)*nc : NULL;
        float mx = -INFINITY;
        for (int64_t i = 0; i < nc; ++i) { float v = xr[i]*scale; if (mr) v += 1.0f*mr[i]; yr[i] = v; if (v > mx) mx = v; }
        double sum = 0.0;
        for (int64_t i = 0; i < nc; ++i) { const float e = expf(yr[i] - mx); yr[i] = e; sum += (double) e; }
        const float inv = (float)(1.0/sum);
        for (int64_t i = 0; i < nc; ++i) yr[i] *= inv;
    }
}
void orc_silu(const float * x, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = x[i]/(1.0f + expf(-x[i])); }   /* :1897 */
void orc_relu(const float * x, float * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : 0.f; }
void orc_add(const float * a, const float * b, float * y, int64_t n, int64_t nb) { for (int64_t i = 0; i < n; ++i) y[i] = a[i] + b[i % nb]; }
void orc_mul(const float * a, const float * b, float * y, int64_t n, int64_t nb) { for (int64_t i = 0; i < n; ++i) y[i] = a[i] * b[i % nb]; }
void orc_cpy_f32_f16(const float * x, uint16_t * y, int64_t n) { for (int64_t i = 0; i < n; ++i) y[i] = orc_fp32_to_fp16(x[i]); }

/* The unfused attention sub-graph of llm_build_kqv (R/src/llama.cpp:706-828) as the CPU evaluates it:
 *   kq  = mul_mat(K f16 [d, n_kv, H_kv], q f32 [d, T, H])      -- q rounded to f16
 *   p   = soft_max_ext(kq, mask, scale)
 *   kqv = mul_mat(V f16 [n_kv, d, H_kv] (transposed cache), p)  -- p rounded to f16
 * out is [d, H, T] (the permute(0,2,1,3)+cont that follows).  Strides are in ELEMENTS. */
void orc_attention(const float * q, const uint16_t * k, const uint16_t * v, const float * mask, float * out,
                   int d, int T, int H, int H_kv, int n_kv, int64_t k_row_stride, int64_t k_head_stride,
                   int64_t v_row_stride, int64_t v_head_stride, int64_t mask_stride, float scale) {
    const int rep = H / H_kv;
    float * kq = (float *) malloc((size_t) n_kv * T * 4), * p = (float *) malloc((size_t) n_kv * T * 4);
    for (int h = 0; h < H; ++h) {
        const uint16_t * kh = k + (int64_t)(h/rep)*k_head_stride, * vh = v + (int64_t)(h/rep)*v_head_stride;
        /* q for head h: [d, T] with q laid out [d, H, T] before the permute */
        float * qh = (float *) malloc((size_t) d * T * 4);
        for (int t = 0; t < T; ++t) memcpy(qh + t*d, q + ((int64_t) t*H + h)*d, (size_t) d*4);
        orc_mul_mat_f16(kh, k_row_stride, qh, kq, d, n_kv, T);
        /* softmax over n_kv for each t, mask row t */
        for (int t = 0; t < T; ++t) {
            const float * xr = kq + (int64_t) t*n_kv; float * yr = p + (int64_t) t*n_kv; const float * mr = mask + (int64_t) t*mask_stride;
            float mx = -INFINITY;
            for (int i = 0; i < n_kv; ++i) { float z = xr[i]*scale; z += 1.0f*mr[i]; yr[i] = z; if (z > mx) mx = z; }
            double sum = 0.0;
            for (int i = 0; i < n_kv; ++i) { const float e = expf(yr[i] - mx); yr[i] = e; sum += (double) e; }
            const float inv = (float)(1.0/sum);
            for (int i = 0; i < n_kv; ++i) yr[i] *= inv;
        }
        float * o = (float *) malloc((size_t) d * T * 4);
        orc_mul_mat_f16(vh, v_row_stride, p, o, n_kv, d, T);
        for (int t = 0; t < T; ++t) memcpy(out + ((int64_t) t*H + h)*d, o + (int64_t) t*d, (size_t) d*4);
        free(o); free(qh);
    }
    free(kq); free(p);
}
