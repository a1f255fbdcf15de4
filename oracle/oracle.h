/* oracle/oracle.h -- declarations of the CPU restatement (TEST INFRASTRUCTURE; see ref_quants.c / ref_ops.c). */
#ifndef ORACLE_H
#define ORACLE_H
#include <stddef.h>
#include <stdint.h>

enum { ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q8_0 = 8, ORC_Q4_K = 12, ORC_Q5_K = 13, ORC_Q6_K = 14 };  /* = ggml_type values */

#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; }                                   orc_block_q4_0;
typedef struct { uint16_t d; int8_t  qs[32]; }                                   orc_block_q8_0;
typedef struct { uint16_t d, s; int8_t qs[32]; }                                 orc_block_q8_1;
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; }        orc_block_q4_K;
typedef struct { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } orc_block_q5_K;
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; } orc_block_q6_K;
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; }                   orc_block_q8_K;
#pragma pack(pop)

float    orc_fp16_to_fp32(uint16_t h);
uint16_t orc_fp32_to_fp16(float f);
int      orc_type_block(int type);
int      orc_type_size(int type);

void orc_quantize_row_q8_0(const float * x, orc_block_q8_0 * y, int64_t k);
void orc_quantize_row_q8_1(const float * x, orc_block_q8_1 * y, int64_t k);
void orc_quantize_row_q8_K(const float * x, orc_block_q8_K * y, int64_t k);
void orc_quantize_row_q4_0(const float * x, orc_block_q4_0 * y, int64_t k);
void orc_dequantize_row_q4_0(const orc_block_q4_0 * x, float * y, int64_t k);
void orc_dequantize_row_q8_0(const orc_block_q8_0 * x, float * y, int64_t k);
void orc_dequantize_row_q4_K(const orc_block_q4_K * x, float * y, int64_t k);
void orc_dequantize_row_q5_K(const orc_block_q5_K * x, float * y, int64_t k);
void orc_dequantize_row_q6_K(const orc_block_q6_K * x, float * y, int64_t k);
void orc_dequantize_row(int type, const void * x, float * y, int64_t k);
float orc_vec_dot_q4_0_q8_0(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q8_0_q8_0(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q4_K_q8_K(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y);
float orc_vec_dot_q5_K_q8_K(int64_t n, const orc_block_q5_K * x, const orc_block_q8_K * y);
float orc_vec_dot_q6_K_q8_K(int64_t n, const orc_block_q6_K * x, const orc_block_q8_K * y);

/* ref_ops.c -- whole ops on plain arrays (contiguous, ggml dimension order: ne0 fastest) */
void orc_mul_mat_q(int wtype, const void * w, const float * x, float * dst, int64_t k, int64_t rows, int64_t T);
void orc_mul_mat_f16(const uint16_t * a, int64_t a_row_stride, const float * x, float * dst, int64_t k, int64_t rows, int64_t T);
void orc_mul_mat_f32(const float * a, const float * x, float * dst, int64_t k, int64_t rows, int64_t T);
void orc_rms_norm(const float * x, float * y, int64_t ne0, int64_t nrows, float eps);
void orc_rope(const float * x, const int32_t * pos, float * y, int64_t ne0, int64_t ne1, int64_t ne2, int n_dims, int mode,
              float freq_base, float freq_scale, float ext_factor, float attn_factor, float beta_fast, float beta_slow, int n_ctx_orig);
void orc_soft_max(const float * x, const float * mask, float * y, int64_t nc, int64_t ne01, int64_t ne02, float scale);
void orc_silu(const float * x, float * y, int64_t n);
void orc_relu(const float * x, float * y, int64_t n);
void orc_add(const float * a, const float * b, float * y, int64_t n, int64_t nb);
void orc_mul(const float * a, const float * b, float * y, int64_t n, int64_t nb);
void orc_cpy_f32_f16(const float * x, uint16_t * y, int64_t n);
/* attention exactly as the CPU backend evaluates the unfused graph (llm_build_kqv, R/src/llama.cpp:706-828) */
void orc_attention(const float * q, const uint16_t * k, const uint16_t * v, const float * mask, float * out,
                   int d, int T, int H, int H_kv, int n_kv, int64_t k_row_stride, int64_t k_head_stride,
                   int64_t v_row_stride, int64_t v_head_stride, int64_t mask_stride, float scale);
#endif
