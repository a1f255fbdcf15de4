/*
 * oracle/ref_quants.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of the reference's CPU arithmetic for the quantised formats on the EAGLE hot
 * path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it, and only as
 * the checker.  Parity status: PINNED -- tests/test_oracle.py checks every function here bit-for-bit
 * against the reference itself (oracle/_ref/libggml-ref-scalar.so, the ISA-independent branches,
 * built from /root/reference by oracle/Makefile) and against the committed vectors in tests/golden/
 * that the same reference produced (tests/golden/make_golden.py).
 *
 * R = /root/reference/llama.cpp.  Each function names the lines it follows.
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

/* ---- IEEE binary16 <-> binary32 (R/ggml/src/ggml-impl.h: ggml_compute_fp16_to_fp32 / fp32_to_fp16,
 *      the portable branch; round-to-nearest-even, subnormals kept) ---- */
float orc_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1f;
    const uint32_t man  = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) { bits = sign; }
        else {                                  /* subnormal: normalise */
            int e = -1; uint32_t m = man;
            do { e++; m <<= 1; } while ((m & 0x400u) == 0);
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | ((m & 0x3ffu) << 13);
        }
    } else if (exp == 31) { bits = sign | 0x7f800000u | (man << 13); }
    else { bits = sign | ((exp + 112) << 23) | (man << 13); }
    float f; memcpy(&f, &bits, 4); return f;
}
uint16_t orc_fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    const uint32_t ax = x & 0x7fffffffu;
    if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);           /* NaN */
    const int e = (int)(ax >> 23) - 127;
    if (e > 15) return (uint16_t)(sign | 0x7c00u);                     /* inf / overflow */
    const uint32_t m = (ax & 0x7fffffu) | 0x800000u;                   /* 24-bit significand */
    int shift = 13; uint32_t base = 0;
    if (e < -14) { shift = 13 + (-14 - e); if (shift > 25) return (uint16_t) sign; }
    else base = (uint32_t)(e + 14) << 10;                              /* exponent field minus one: q carries the hidden bit */
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;                  /* nearest, ties to even; carries roll into the exponent */
    return (uint16_t)(sign | (base + q));
}

/* nearest_int, R/ggml/src/ggml-quants.c:559-565 (magic-number rounding, ties to even) */
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

/* ---- activation quantisers ---- */
/* quantize_row_q8_0_ref, R/ggml/src/ggml-quants.c:194-215 */
void orc_quantize_row_q8_0(const float * x, orc_block_q8_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = fabsf(x[i*32 + j]); if (v > amax) amax = v; }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        for (int j = 0; j < 32; ++j) y[i].qs[j] = (int8_t) roundf(x[i*32 + j]*id);
    }
}
/* quantize_row_q8_1_ref, R/ggml/src/ggml-quants.c:220-246 (what the reference GPU path uses instead) */
void orc_quantize_row_q8_1(const float * x, orc_block_q8_1 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = fabsf(x[i*32 + j]); if (v > amax) amax = v; }
        const float d = amax / ((1 << 7) - 1);
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        int sum = 0;
        for (int j = 0; j < 16; ++j) {
            const float v0 = x[i*32 + j]*id, v1 = x[i*32 + 16 + j]*id;
            y[i].qs[j] = (int8_t) roundf(v0); y[i].qs[16 + j] = (int8_t) roundf(v1);
            sum += y[i].qs[j]; sum += y[i].qs[16 + j];
        }
        y[i].s = orc_fp32_to_fp16(sum*d);
    }
}
/* quantize_row_q8_K_ref, R/ggml/src/ggml-quants.c:2479-2512 */
void orc_quantize_row_q8_K(const float * x, orc_block_q8_K * y, int64_t k) {
    const int64_t nb = k / 256;
    for (int64_t i = 0; i < nb; i++) {
        float max = 0, amax = 0;
        for (int j = 0; j < 256; ++j) { const float ax = fabsf(x[j]); if (ax > amax) { amax = ax; max = x[j]; } }
        if (!amax) { y[i].d = 0; memset(y[i].qs, 0, 256); memset(y[i].bsums, 0, sizeof(y[i].bsums)); x += 256; continue; }
        const float iscale = -127.f/max;
        for (int j = 0; j < 256; ++j) { int v = nearest_int(iscale*x[j]); y[i].qs[j] = (int8_t)(v < 127 ? v : 127); }
        for (int j = 0; j < 16; ++j) { int sum = 0; for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j*16 + ii]; y[i].bsums[j] = (int16_t) sum; }
        y[i].d = 1/iscale;
        x += 256;
    }
}

/* ---- weight-side helpers ---- */
/* quantize_row_q4_0_ref / q8_0 weights, R/ggml/src/ggml-quants.c:37-70,194-215 (used to make test weights) */
void orc_quantize_row_q4_0(const float * x, orc_block_q4_0 * y, int64_t k) {
    const int64_t nb = k / 32;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f, max = 0.0f;
        for (int j = 0; j < 32; j++) { const float v = x[i*32 + j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
        const float d = max / -8;
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        for (int j = 0; j < 16; ++j) {
            const float x0 = x[i*32 + j]*id, x1 = x[i*32 + 16 + j]*id;
            int a = (int8_t)(x0 + 8.5f), b = (int8_t)(x1 + 8.5f);
            const uint8_t xi0 = (uint8_t)(a < 15 ? a : 15), xi1 = (uint8_t)(b < 15 ? b : 15);
            y[i].qs[j] = (uint8_t)(xi0 | (xi1 << 4));
        }
    }
}
/* get_scale_min_k4, R/ggml/src/ggml-quants.c:631-638 */
static inline void scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else { *d = (uint8_t)((q[j+4] & 0xF) | ((q[j-4] >> 6) << 4)); *m = (uint8_t)((q[j+4] >> 4) | ((q[j] >> 6) << 4)); }
}

/* dequantize_row_q4_0 :255, q8_0 :349, q4_K :1280, q5_K :1482, q6_K :1690 of R/ggml/src/ggml-quants.c */
void orc_dequantize_row_q4_0(const orc_block_q4_0 * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k/32; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int j = 0; j < 16; ++j) { y[i*32 + j] = ((x[i].qs[j] & 0x0F) - 8)*d; y[i*32 + j + 16] = ((x[i].qs[j] >> 4) - 8)*d; }
    }
}
void orc_dequantize_row_q8_0(const orc_block_q8_0 * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k/32; i++) { const float d = orc_fp16_to_fp32(x[i].d); for (int j = 0; j < 32; ++j) y[i*32 + j] = x[i].qs[j]*d; }
}
void orc_dequantize_row_q4_K(const orc_block_q4_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k/256; i++) {
        const uint8_t * q = x[i].qs;
        const float d = orc_fp16_to_fp32(x[i].d), min = orc_fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m;
        for (int j = 0; j < 256; j += 64) {
            scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d*sc, m1 = min*m;
            scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d*sc, m2 = min*m;
            for (int l = 0; l < 32; ++l) *y++ = d1*(q[l] & 0xF) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2*(q[l] >> 4) - m2;
            q += 32; is += 2;
        }
    }
}
void orc_dequantize_row_q5_K(const orc_block_q5_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k/256; i++) {
        const uint8_t * ql = x[i].qs, * qh = x[i].qh;
        const float d = orc_fp16_to_fp32(x[i].d), min = orc_fp16_to_fp32(x[i].dmin);
        int is = 0; uint8_t sc, m, u1 = 1, u2 = 2;
        for (int j = 0; j < 256; j += 64) {
            scale_min_k4(is + 0, x[i].scales, &sc, &m); const float d1 = d*sc, m1 = min*m;
            scale_min_k4(is + 1, x[i].scales, &sc, &m); const float d2 = d*sc, m2 = min*m;
            for (int l = 0; l < 32; ++l) *y++ = d1*((ql[l] & 0xF) + (qh[l] & u1 ? 16 : 0)) - m1;
            for (int l = 0; l < 32; ++l) *y++ = d2*((ql[l] >> 4) + (qh[l] & u2 ? 16 : 0)) - m2;
            ql += 32; is += 2; u1 <<= 2; u2 <<= 2;
        }
    }
}
void orc_dequantize_row_q6_K(const orc_block_q6_K * x, float * y, int64_t k) {
    for (int64_t i = 0; i < k/256; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        const uint8_t * ql = x[i].ql, * qh = x[i].qh; const int8_t * sc = x[i].scales;
        for (int n = 0; n < 256; n += 128) {
            for (int l = 0; l < 32; ++l) {
                const int is = l/16;
                const int8_t q1 = (int8_t)((ql[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l +  0] >>  4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32] >>  4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l +  0] = d*sc[is + 0]*q1; y[l + 32] = d*sc[is + 2]*q2; y[l + 64] = d*sc[is + 4]*q3; y[l + 96] = d*sc[is + 6]*q4;
            }
            y += 128; ql += 64; qh += 32; sc += 8;
        }
    }
}

/* ---- dot products, ISA-independent branches of R/ggml/src/ggml-cpu/ggml-cpu-quants.c ---- */
/* ggml_vec_dot_q4_0_q8_0 scalar tail :2592-2607 */
float orc_vec_dot_q4_0_q8_0(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n/32; ++ib) {
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; ++j) { s0 += ((x[ib].qs[j] & 0x0F) - 8)*y[ib].qs[j]; s1 += ((x[ib].qs[j] >> 4) - 8)*y[ib].qs[j + 16]; }
        const int sumi = s0 + s1;
        sumf += sumi*orc_fp16_to_fp32(x[ib].d)*orc_fp16_to_fp32(y[ib].d);
    }
    return sumf;
}
/* ggml_vec_dot_q8_0_q8_0 scalar tail :4068-4078 */
float orc_vec_dot_q8_0_q8_0(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y) {
    float sumf = 0;
    for (int64_t ib = 0; ib < n/32; ++ib) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += x[ib].qs[j]*y[ib].qs[j];
        sumf += sumi*(orc_fp16_to_fp32(x[ib].d)*orc_fp16_to_fp32(y[ib].d));
    }
    return sumf;
}
/* shared tail of the K-quant dots: 8 interleaved int32 lanes per super-block, lane l collects elements
 * with index = l (mod 8); per block sums[l] += d*aux32[l]; the 8 float lanes are added at the very end. */
static inline void lanes_add(int32_t * aux32, int scale, const int8_t * q8, const int8_t * a, int n8) {
    for (int g = 0; g < n8; ++g) for (int l = 0; l < 8; ++l) aux32[l] += scale * (int16_t)(q8[g*8 + l] * a[g*8 + l]);
}
/* ggml_vec_dot_q4_K_q8_K, scalar branch :7020-7078 */
float orc_vec_dot_q4_K_q8_K(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y) {
    float sums[8] = {0}, sumf = 0;
    for (int64_t i = 0; i < n/256; ++i) {
        int8_t a[256]; int32_t aux32[8] = {0};
        for (int j = 0; j < 4; ++j) for (int l = 0; l < 32; ++l) { a[64*j + l] = (int8_t)(x[i].qs[32*j + l] & 0xF); a[64*j + 32 + l] = (int8_t)(x[i].qs[32*j + l] >> 4); }
        int sumi = 0;
        for (int j = 0; j < 16; ++j) { uint8_t sc, m; scale_min_k4(j/2, x[i].scales, &sc, &m); sumi += y[i].bsums[j]*m; }
        for (int j = 0; j < 8; ++j) { uint8_t sc, m; scale_min_k4(j, x[i].scales, &sc, &m); lanes_add(aux32, sc, y[i].qs + 32*j, a + 32*j, 4); }
        const float d = orc_fp16_to_fp32(x[i].d)*y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d*aux32[l];
        const float dmin = orc_fp16_to_fp32(x[i].dmin)*y[i].d;
        sumf -= dmin*sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ggml_vec_dot_q5_K_q8_K, scalar `#else` tail :7841-7901 */
float orc_vec_dot_q5_K_q8_K(int64_t n, const orc_block_q5_K * x, const orc_block_q8_K * y) {
    float sums[8] = {0}, sumf = 0;
    for (int64_t i = 0; i < n/256; ++i) {
        int8_t a[256]; int32_t aux32[8] = {0};
        for (int j = 0; j < 4; ++j) for (int l = 0; l < 32; ++l) {
            a[64*j + l]      = (int8_t)((x[i].qs[32*j + l] & 0xF) + ((x[i].qh[l] >> (2*j))     & 1 ? 16 : 0));
            a[64*j + 32 + l] = (int8_t)((x[i].qs[32*j + l] >> 4)  + ((x[i].qh[l] >> (2*j + 1)) & 1 ? 16 : 0));
        }
        int sumi = 0;
        for (int j = 0; j < 16; ++j) { uint8_t sc, m; scale_min_k4(j/2, x[i].scales, &sc, &m); sumi += y[i].bsums[j]*m; }
        for (int j = 0; j < 8; ++j) { uint8_t sc, m; scale_min_k4(j, x[i].scales, &sc, &m); lanes_add(aux32, sc, y[i].qs + 32*j, a + 32*j, 4); }
        const float d = orc_fp16_to_fp32(x[i].d)*y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d*aux32[l];
        const float dmin = orc_fp16_to_fp32(x[i].dmin)*y[i].d;
        sumf -= dmin*sumi;
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}
/* ggml_vec_dot_q6_K_q8_K, scalar `#else` tail :8679-8720 */
float orc_vec_dot_q6_K_q8_K(int64_t n, const orc_block_q6_K * x, const orc_block_q8_K * y) {
    float sums[8] = {0}, sumf = 0;
    for (int64_t i = 0; i < n/256; ++i) {
        int8_t a[256]; int32_t aux32[8] = {0};
        for (int h = 0; h < 2; ++h) for (int l = 0; l < 32; ++l) {
            const uint8_t * ql = x[i].ql + 64*h, * qh = x[i].qh + 32*h;
            a[128*h + l +  0] = (int8_t)((ql[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
            a[128*h + l + 32] = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
            a[128*h + l + 64] = (int8_t)((ql[l +  0] >>  4) | (((qh[l] >> 4) & 3) << 4)) - 32;
            a[128*h + l + 96] = (int8_t)((ql[l + 32] >>  4) | (((qh[l] >> 6) & 3) << 4)) - 32;
        }
        for (int j = 0; j < 16; ++j) lanes_add(aux32, x[i].scales[j], y[i].qs + 16*j, a + 16*j, 2);
        const float d = orc_fp16_to_fp32(x[i].d)*y[i].d;
        for (int l = 0; l < 8; ++l) sums[l] += d*aux32[l];
    }
    for (int l = 0; l < 8; ++l) sumf += sums[l];
    return sumf;
}

/* sizes */
int orc_type_block(int type) { switch (type) { case ORC_Q4_0: case ORC_Q8_0: return 32; case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: return 256; case ORC_F32: case ORC_F16: return 1; default: return 0; } }
int orc_type_size(int type)  { switch (type) { case ORC_Q4_0: return 18; case ORC_Q8_0: return 34; case ORC_Q4_K: return 144; case ORC_Q5_K: return 176; case ORC_Q6_K: return 210; case ORC_F32: return 4; case ORC_F16: return 2; default: return 0; } }

void orc_dequantize_row(int type, const void * x, float * y, int64_t k) {
    switch (type) {
        case ORC_Q4_0: orc_dequantize_row_q4_0((const orc_block_q4_0 *) x, y, k); break;
        case ORC_Q8_0: orc_dequantize_row_q8_0((const orc_block_q8_0 *) x, y, k); break;
        case ORC_Q4_K: orc_dequantize_row_q4_K((const orc_block_q4_K *) x, y, k); break;
        case ORC_Q5_K: orc_dequantize_row_q5_K((const orc_block_q5_K *) x, y, k); break;
        case ORC_Q6_K: orc_dequantize_row_q6_K((const orc_block_q6_K *) x, y, k); break;
        case ORC_F32:  memcpy(y, x, (size_t) k*4); break;
        case ORC_F16:  for (int64_t i = 0; i < k; ++i) y[i] = orc_fp16_to_fp32(((const uint16_t *) x)[i]); break;
        default: break;
    }
}
