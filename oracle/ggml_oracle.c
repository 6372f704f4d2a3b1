/*
 * oracle/ggml_oracle.c -- TEST INFRASTRUCTURE.  NOT PART OF THE PRODUCT PATH.
 *
 * CPU restatement (plain C11, no SIMD) of the reference's algorithm for the transformer
 * forward hot path.  See ggml_oracle.h for the parity pin and the per-function citations.
 * Floating point: compiled with -ffp-contract=off so that every fmaf below is intentional.
 */
#include "ggml_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* two statements of the reference are plain C that gcc -O3 (-ffp-contract=fast) compiles to one fma; fused here explicitly */
#define ORC_Q41_SUMMS(m, s, acc) fmaf((m), (s), (acc))
#ifndef ORC_CODEBOOK_TAIL
#define ORC_CODEBOOK_TAIL(d, v, acc) ((acc) + (d) * (v))          /* `sumf += d * (sumi1 + sumi2)` of the IQ4_NL / MXFP4 tail, as compiled in the reference build: two roundings */
#endif
#ifndef ORC_F32_TAIL_VW
#define ORC_F32_TAIL_VW 4
#endif
#define ORC_MIN(a, b) ((a) < (b) ? (a) : (b))
#define ORC_MAX(a, b) ((a) > (b) ? (a) : (b))

/* ------------------------------------------------------------------------------------------ */
/* type traits                                                                                 */
/* ------------------------------------------------------------------------------------------ */
size_t orc_type_size(int type) {
    switch (type) {
        case ORC_F32: return 4;  case ORC_F16: return 2;  case ORC_I32: return 4;  case ORC_I64: return 8;
        case ORC_Q4_0: return sizeof(orc_block_q4_0);  case ORC_Q8_0: return sizeof(orc_block_q8_0);
        case ORC_Q4_1: return sizeof(orc_block_q4_1);  case ORC_Q8_1: return sizeof(orc_block_q8_1);
        case ORC_Q4_K: return sizeof(orc_block_q4_K);  case ORC_Q8_K: return sizeof(orc_block_q8_K);
        case ORC_Q5_K: return sizeof(orc_block_q5_K);  case ORC_Q6_K: return sizeof(orc_block_q6_K);
        case ORC_Q5_0: return sizeof(orc_block_q5_0);  case ORC_Q5_1: return sizeof(orc_block_q5_1);
        case ORC_IQ4_NL: return sizeof(orc_block_iq4_nl);  case ORC_MXFP4: return sizeof(orc_block_mxfp4);  case ORC_IQ4_XS: return sizeof(orc_block_iq4_xs);
        case ORC_TQ1_0: return sizeof(orc_block_tq1_0);  case ORC_TQ2_0: return sizeof(orc_block_tq2_0);
        case ORC_IQ2_XXS: return sizeof(orc_block_iq2_xxs);  case ORC_IQ2_XS: return sizeof(orc_block_iq2_xs);  case ORC_IQ2_S: return sizeof(orc_block_iq2_s);
        case ORC_IQ3_XXS: return sizeof(orc_block_iq3_xxs);  case ORC_IQ3_S: return sizeof(orc_block_iq3_s);
        case ORC_IQ1_S: return sizeof(orc_block_iq1_s);  case ORC_IQ1_M: return sizeof(orc_block_iq1_m);
        case ORC_Q2_K: return sizeof(orc_block_q2_K);  case ORC_Q3_K: return sizeof(orc_block_q3_K);
    }
    return 0;
}
int orc_blck_size(int type) {
    switch (type) {
        case ORC_Q4_0: case ORC_Q8_0: case ORC_Q4_1: case ORC_Q8_1: case ORC_Q5_0: case ORC_Q5_1: case ORC_IQ4_NL: case ORC_MXFP4: return ORC_QK;
        case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: case ORC_Q8_K: case ORC_Q2_K: case ORC_Q3_K: case ORC_IQ4_XS: case ORC_TQ1_0: case ORC_TQ2_0: case ORC_IQ2_XXS: case ORC_IQ2_XS: case ORC_IQ2_S: case ORC_IQ3_XXS: case ORC_IQ3_S: case ORC_IQ1_S: case ORC_IQ1_M: return ORC_QK_K;
        case ORC_F32: case ORC_F16: case ORC_I32: case ORC_I64: return 1;
    }
    return 0;
}
size_t orc_row_size(int type, int64_t ne) { return orc_type_size(type) * (size_t)(ne / orc_blck_size(type)); }

/* ------------------------------------------------------------------------------------------ */
/* fp16 (IEEE binary16, round-to-nearest-even: the behaviour of F16C vcvtps2ph / vcvtph2ps)    */
/* ------------------------------------------------------------------------------------------ */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float orc_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1f;
    const uint32_t man  = h & 0x3ffu;
    if (exp == 0) {
        if (man == 0) return u2f(sign);
        /* subnormal: value = man * 2^-24 (exact in f32) */
        float v = (float) man * 0x1p-24f;
        return u2f(f2u(v) | sign);
    }
    if (exp == 31) return u2f(sign | 0x7f800000u | (man << 13));
    return u2f(sign | ((exp + 112u) << 23) | (man << 13));
}

uint16_t orc_fp32_to_fp16(float f) {
    const uint32_t x    = f2u(f);
    const uint16_t sign = (uint16_t)((x >> 16) & 0x8000u);
    const uint32_t ax   = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) {                       /* inf / nan */
        return (uint16_t)(sign | 0x7c00u | (ax > 0x7f800000u ? (0x200u | ((ax >> 13) & 0x3ffu)) : 0));
    }
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);  /* >= 65520 rounds to inf */
    if (ax < 0x33000001u) return sign;                          /* <= 2^-25 rounds to zero (tie to even) */
    int32_t  e = (int32_t)(ax >> 23) - 127;        /* unbiased exponent */
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;     /* 24-bit significand */
    int shift;
    uint32_t base;
    if (e < -14) { shift = 13 + (-14 - e); base = 0; }           /* subnormal half */
    else         { shift = 13;             base = (uint32_t)(e + 15) << 10; m &= 0x7fffffu; }
    uint32_t q    = m >> shift;
    uint32_t rem  = m & ((1u << shift) - 1u);
    uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q));          /* carry into the exponent is the correct rounding */
}

/* ------------------------------------------------------------------------------------------ */
/* activation quantizers                                                                       */
/* ------------------------------------------------------------------------------------------ */
/* arch/x86/quants.c:290-386 (AVX2): d = amax/127 stored as fp16; q = cvt(round_nearest_even(x * (127/amax))) */
void orc_quantize_row_q8_0(const float * x, orc_block_q8_0 * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < ORC_QK; j++) amax = fmaxf(amax, fabsf(x[i*ORC_QK + j]));
        const float d  = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        for (int j = 0; j < ORC_QK; j++) {
            /* nearbyintf under the default rounding mode == _mm256_round_ps(NEAREST) == half-to-even */
            y[i].qs[j] = (int8_t)(int) nearbyintf(x[i*ORC_QK + j] * id);
        }
    }
}

/* arch/x86/quants.c:388-480 (AVX2): the Q8_0 arithmetic, plus s = fp16(d * sum q) where d is the fp32 quotient (not its fp16 rounding) */
void orc_quantize_row_q8_1(const float * x, orc_block_q8_1 * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < ORC_QK; j++) amax = fmaxf(amax, fabsf(x[i*ORC_QK + j]));
        const float d  = amax / 127.f;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        int sum = 0;
        for (int j = 0; j < ORC_QK; j++) {
            const int q = (int) nearbyintf(x[i*ORC_QK + j] * id);
            y[i].qs[j] = (int8_t) q;
            sum += q;
        }
        y[i].s = orc_fp32_to_fp16(d * (float) sum);
    }
}

/* ggml-quants.c:199-222: id = 1/d, roundf (half away from zero) */
void orc_quantize_row_q8_0_ref(const float * x, orc_block_q8_0 * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        float amax = 0.0f;
        for (int j = 0; j < ORC_QK; j++) amax = ORC_MAX(amax, fabsf(x[i*ORC_QK + j]));
        const float d  = amax / 127;
        const float id = d ? 1.0f/d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        for (int j = 0; j < ORC_QK; j++) y[i].qs[j] = (int8_t) roundf(x[i*ORC_QK + j] * id);
    }
}

/* ggml-quants.c:436-441 */
static inline int orc_nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

/* ggml-quants.c:2555-2592 */
void orc_quantize_row_q8_K(const float * x, orc_block_q8_K * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++, x += ORC_QK_K) {
        float max = 0, amax = 0;
        for (int j = 0; j < ORC_QK_K; j++) {
            const float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }   /* first element of largest magnitude, signed */
        }
        if (!amax) {
            /* the reference leaves bsums untouched here; every consumer multiplies them by d == 0.
             * We zero them so that the restatement is deterministic. */
            memset(&y[i], 0, sizeof(y[i]));
            continue;
        }
        const float iscale = -127.f / max;
        for (int j = 0; j < ORC_QK_K; j++) {
            const int v = orc_nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t) ORC_MIN(127, v);
        }
        for (int j = 0; j < ORC_QK_K/16; j++) {
            int sum = 0;
            for (int ii = 0; ii < 16; ii++) sum += y[i].qs[j*16 + ii];
            y[i].bsums[j] = (int16_t) sum;
        }
        y[i].d = 1 / iscale;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* weight dequantizers                                                                         */
/* ------------------------------------------------------------------------------------------ */
void orc_dequantize_row_q4_0(const orc_block_q4_0 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int j = 0; j < 16; j++) {
            y[i*32 + j]      = (float)((x[i].qs[j] & 0x0F) - 8) * d;
            y[i*32 + j + 16] = (float)((x[i].qs[j] >>   4) - 8) * d;
        }
    }
}
void orc_dequantize_row_q4_1(const orc_block_q4_1 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d), m = orc_fp16_to_fp32(x[i].m);
        for (int j = 0; j < 16; j++) {
            y[i*32 + j]      = (float)(x[i].qs[j] & 0x0F) * d + m;
            y[i*32 + j + 16] = (float)(x[i].qs[j] >>   4) * d + m;
        }
    }
}
void orc_dequantize_row_q8_0(const orc_block_q8_0 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int j = 0; j < 32; j++) y[i*32 + j] = (float) x[i].qs[j] * d;
    }
}
/* 6-bit scale/min unpack, ggml-quants.c:703-711 */
static inline void orc_scale_min_k4(int j, const uint8_t * q, uint8_t * sc, uint8_t * m) {
    if (j < 4) { *sc = q[j] & 63;  *m = q[j + 4] & 63; }
    else       { *sc = (uint8_t)((q[j+4] & 0xF) | ((q[j-4] >> 6) << 4));
                 *m  = (uint8_t)((q[j+4] >>  4) | ((q[j  ] >> 6) << 4)); }
}
void orc_dequantize_row_q4_K(const orc_block_q4_K * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d), dmin = orc_fp16_to_fp32(x[i].dmin);
        const uint8_t * q = x[i].qs;
        for (int g = 0; g < 4; g++, q += 32) {     /* 64 weights per group: low nibbles then high nibbles */
            uint8_t sc, m;
            orc_scale_min_k4(2*g + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc, m1 = dmin * m;
            orc_scale_min_k4(2*g + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc, m2 = dmin * m;
            for (int l = 0; l < 32; l++) *y++ = d1 * (float)(q[l] & 0xF) - m1;
            for (int l = 0; l < 32; l++) *y++ = d2 * (float)(q[l] >>  4) - m2;
        }
    }
}
void orc_dequantize_row_q5_K(const orc_block_q5_K * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d), dmin = orc_fp16_to_fp32(x[i].dmin);
        const uint8_t * ql = x[i].qs, * qh = x[i].qh;
        uint8_t u1 = 1, u2 = 2;
        for (int g = 0; g < 4; g++, ql += 32, u1 <<= 2, u2 <<= 2) {
            uint8_t sc, m;
            orc_scale_min_k4(2*g + 0, x[i].scales, &sc, &m);
            const float d1 = d * sc, m1 = dmin * m;
            orc_scale_min_k4(2*g + 1, x[i].scales, &sc, &m);
            const float d2 = d * sc, m2 = dmin * m;
            for (int l = 0; l < 32; l++) *y++ = d1 * (float)((ql[l] & 0xF) + ((qh[l] & u1) ? 16 : 0)) - m1;
            for (int l = 0; l < 32; l++) *y++ = d2 * (float)((ql[l] >>  4) + ((qh[l] & u2) ? 16 : 0)) - m2;
        }
    }
}
void orc_dequantize_row_q6_K(const orc_block_q6_K * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        const uint8_t * ql = x[i].ql, * qh = x[i].qh;
        const int8_t * sc = x[i].scales;
        for (int n = 0; n < ORC_QK_K; n += 128, y += 128, ql += 64, qh += 32, sc += 8)
            for (int l = 0; l < 32; l++) {
                const int is = l / 16;
                const int8_t q1 = (int8_t)((ql[l +  0] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32;
                const int8_t q2 = (int8_t)((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32;
                const int8_t q3 = (int8_t)((ql[l +  0]  >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32;
                const int8_t q4 = (int8_t)((ql[l + 32]  >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32;
                y[l +  0] = d * sc[is + 0] * q1;
                y[l + 32] = d * sc[is + 2] * q2;
                y[l + 64] = d * sc[is + 4] * q3;
                y[l + 96] = d * sc[is + 6] * q4;
            }
    }
}

/* ---- the other formats stock model files carry ---- */
static const int8_t orc_kvalues_iq4nl[16] = { -127, -104, -83, -65, -49, -35, -22, -10, 1, 13, 25, 38, 53, 69, 89, 113 };   /* ggml-common.h:1088-1090 */
static const int8_t orc_kvalues_mxfp4[16] = { 0, 1, 2, 3, 4, 6, 8, 12, 0, -1, -2, -3, -4, -6, -8, -12 };                    /* ggml-common.h:1094-1096 */
/* ggml_e8m0_to_fp32_half (ggml-impl.h:471-489): 2^(e - 128), the two smallest as denormal patterns */
static inline float orc_e8m0_half(uint8_t e) { return u2f(e < 2 ? 0x00200000u << e : (uint32_t)(e - 1) << 23); }
static inline uint32_t orc_qh32(const uint8_t * qh) { uint32_t v; memcpy(&v, qh, 4); return v; }
/* element e (0..31) of a 5-bit block: nibble | fifth bit << 4 */
static inline int orc_q5_elem(const uint8_t * qs, uint32_t qh, int e) {
    const int nib = e < 16 ? (qs[e] & 0x0F) : (qs[e - 16] >> 4);
    return nib | (int)(((qh >> e) & 1u) << 4);
}
static inline int orc_nib_elem(const uint8_t * qs, int e) { return e < 16 ? (qs[e] & 0x0F) : (qs[e - 16] >> 4); }

void orc_dequantize_row_q5_0(const orc_block_q5_0 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        const uint32_t qh = orc_qh32(x[i].qh);
        for (int e = 0; e < 32; e++) y[i*32 + e] = (float)(orc_q5_elem(x[i].qs, qh, e) - 16) * d;
    }
}
void orc_dequantize_row_q5_1(const orc_block_q5_1 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d), m = orc_fp16_to_fp32(x[i].m);
        const uint32_t qh = orc_qh32(x[i].qh);
        for (int e = 0; e < 32; e++) y[i*32 + e] = (float) orc_q5_elem(x[i].qs, qh, e) * d + m;
    }
}
void orc_dequantize_row_mxfp4(const orc_block_mxfp4 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_e8m0_half(x[i].e);
        for (int e = 0; e < 32; e++) y[i*32 + e] = (float) orc_kvalues_mxfp4[orc_nib_elem(x[i].qs, e)] * d;
    }
}
void orc_dequantize_row_iq4_nl(const orc_block_iq4_nl * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int e = 0; e < 32; e++) y[i*32 + e] = d * (float) orc_kvalues_iq4nl[orc_nib_elem(x[i].qs, e)];
    }
}
void orc_dequantize_row_iq4_xs(const orc_block_iq4_xs * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int ib = 0; ib < 8; ib++) {
            const int ls = ((x[i].scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((x[i].scales_h >> (2 * ib)) & 3) << 4);
            const float dl = d * (float)(ls - 32);
            for (int e = 0; e < 32; e++) y[i*256 + ib*32 + e] = dl * (float) orc_kvalues_iq4nl[orc_nib_elem(x[i].qs + 16 * ib, e)];
        }
    }
}
/* ---- the grid formats.  One decoder per format: the 256 signed codebook magnitudes of a super-block (int8), the odd integer scale of every 16 elements (the AVX2 dot
 * products multiply by it) and the float the dequantizer multiplies by (db, in ITS operation order) ---- */
#include "../chatllm.cpp_amd/csrc/iq_grids.h"
typedef struct { int8_t w[256]; int sc16[16]; float db16[16]; float d; } orc_iq_dec;
static void orc_iq_put8(orc_iq_dec * o, int e0, uint32_t lo4, uint32_t hi4, uint32_t signs8) {
    const uint32_t a = iq_apply_signs4(lo4, signs8 & 15u), b = iq_apply_signs4(hi4, signs8 >> 4);
    for (int k = 0; k < 4; k++) { o->w[e0 + k] = (int8_t)((a >> (8 * k)) & 0xff); o->w[e0 + 4 + k] = (int8_t)((b >> (8 * k)) & 0xff); }
}
static void orc_iq_decode(int type, const void * blk, orc_iq_dec * o) {
    if (type == ORC_IQ2_XXS) {                      /* dequantize_row_iq2_xxs (ggml-quants.c:2275-2303): db = d * (0.5f + ls) * 0.25f */
        const orc_block_iq2_xxs * b = (const orc_block_iq2_xxs *) blk;
        o->d = orc_fp16_to_fp32(b->d);
        for (int ib = 0; ib < 8; ib++) {
            uint32_t aux[2]; memcpy(aux, b->qs + 4 * ib, 8);
            const int ls = (int)(aux[1] >> 28);
            for (int h = 0; h < 2; h++) { o->sc16[2 * ib + h] = 2 * ls + 1; o->db16[2 * ib + h] = o->d * (0.5f + (float) ls) * 0.25f; }
            for (int l = 0; l < 4; l++) {
                const uint32_t code = IQ2XXS_CODE[(aux[0] >> (8 * l)) & 0xff];
                orc_iq_put8(o, 32 * ib + 8 * l, iq2_code_bytes4(code, 0), iq2_code_bytes4(code, 4), iq_ksigns((aux[1] >> (7 * l)) & 127));
            }
        }
    } else if (type == ORC_IQ2_XS) {                /* dequantize_row_iq2_xs (:2307-2334): a 4-bit scale per 16 */
        const orc_block_iq2_xs * b = (const orc_block_iq2_xs *) blk;
        o->d = orc_fp16_to_fp32(b->d);
        for (int ib = 0; ib < 8; ib++) {
            for (int h = 0; h < 2; h++) { const int ls = (b->scales[ib] >> (4 * h)) & 0xf; o->sc16[2 * ib + h] = 2 * ls + 1; o->db16[2 * ib + h] = o->d * (0.5f + (float) ls) * 0.25f; }
            for (int l = 0; l < 4; l++) {
                const uint32_t q = b->qs[4 * ib + l], code = IQ2XS_CODE[q & 511];
                orc_iq_put8(o, 32 * ib + 8 * l, iq2_code_bytes4(code, 0), iq2_code_bytes4(code, 4), iq_ksigns(q >> 9));
            }
        }
    } else if (type == ORC_IQ2_S) {                 /* dequantize_row_iq2_s (:2338-2371): 10-bit grid index (qs | two bits of qh), explicit sign bytes */
        const orc_block_iq2_s * b = (const orc_block_iq2_s *) blk;
        o->d = orc_fp16_to_fp32(b->d);
        for (int ib = 0; ib < 8; ib++) {
            for (int h = 0; h < 2; h++) { const int ls = (b->scales[ib] >> (4 * h)) & 0xf; o->sc16[2 * ib + h] = 2 * ls + 1; o->db16[2 * ib + h] = o->d * (0.5f + (float) ls) * 0.25f; }
            for (int l = 0; l < 4; l++) {
                const uint32_t code = IQ2S_CODE[b->qs[4 * ib + l] | ((b->qh[ib] << (8 - 2 * l)) & 0x300)];
                orc_iq_put8(o, 32 * ib + 8 * l, iq2_code_bytes4(code, 0), iq2_code_bytes4(code, 4), b->qs[32 + 4 * ib + l]);
            }
        }
    } else if (type == ORC_IQ3_XXS) {               /* dequantize_row_iq3_xxs (:2375-2407): db = d * (0.5f + ls) * 0.5f */
        const orc_block_iq3_xxs * b = (const orc_block_iq3_xxs *) blk;
        o->d = orc_fp16_to_fp32(b->d);
        for (int ib = 0; ib < 8; ib++) {
            uint32_t aux; memcpy(&aux, b->qs + 64 + 4 * ib, 4);
            const int ls = (int)(aux >> 28);
            for (int h = 0; h < 2; h++) { o->sc16[2 * ib + h] = 2 * ls + 1; o->db16[2 * ib + h] = o->d * (0.5f + (float) ls) * 0.5f; }
            for (int l = 0; l < 4; l++)
                orc_iq_put8(o, 32 * ib + 8 * l, iq3xxs_code_bytes4(IQ3XXS_CODE[b->qs[8 * ib + 2 * l]]), iq3xxs_code_bytes4(IQ3XXS_CODE[b->qs[8 * ib + 2 * l + 1]]), iq_ksigns((aux >> (7 * l)) & 127));
        }
    } else {                                        /* dequantize_row_iq3_s (:2411-2460): db = d * (1 + 2 * ls), one 4-bit scale per 32 */
        const orc_block_iq3_s * b = (const orc_block_iq3_s *) blk;
        o->d = orc_fp16_to_fp32(b->d);
        for (int ib = 0; ib < 8; ib++) {
            const int ls = (b->scales[ib / 2] >> (4 * (ib & 1))) & 0xf;
            for (int h = 0; h < 2; h++) { o->sc16[2 * ib + h] = 2 * ls + 1; o->db16[2 * ib + h] = o->d * (float)(1 + 2 * ls); }
            for (int l = 0; l < 4; l++) {
                const uint32_t i1 = b->qs[8 * ib + 2 * l] | ((b->qh[ib] << (8 - 2 * l)) & 256), i2 = b->qs[8 * ib + 2 * l + 1] | ((b->qh[ib] << (7 - 2 * l)) & 256);
                orc_iq_put8(o, 32 * ib + 8 * l, iq3s_code_bytes4(IQ3S_CODE[i1]), iq3s_code_bytes4(IQ3S_CODE[i2]), b->signs[4 * ib + l]);
            }
        }
    }
}
void orc_dequantize_row_iq_grid(int type, const void * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K; const size_t bs = orc_type_size(type);
    orc_iq_dec o;
    for (int64_t i = 0; i < nb; i++) {
        orc_iq_decode(type, (const char *) x + i * bs, &o);
        for (int e = 0; e < 256; e++) {             /* y = db * grid[j] * (sign ? -1.f : 1.f) */
            const int m = o.w[e] < 0 ? -o.w[e] : o.w[e];
            y[i*256 + e] = o.db16[e / 16] * (float) m * (o.w[e] < 0 ? -1.0f : 1.0f);
        }
    }
}
static inline float hsum8(const float x[8]);
float orc_vec_dot_iq_grid_q8_K_avx2(int type, int64_t n, const void * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K; const size_t bs = orc_type_size(type);
    float acc[8] = {0};
    orc_iq_dec o;
    for (int64_t i = 0; i < nb; i++) {
        orc_iq_decode(type, (const char *) x + i * bs, &o);
        const float d = o.d * y[i].d;
        for (int L = 0; L < 8; L++) {
            int32_t sumi = 0;
            for (int c = 0; c < 8; c++) {
                int p = 0;
                for (int e = 0; e < 4; e++) p += (int) o.w[32 * c + 4 * L + e] * (int) y[i].qs[32 * c + 4 * L + e];
                sumi += o.sc16[2 * c + (L >> 2)] * p;
            }
            acc[L] = fmaf(d, (float) sumi, acc[L]);
        }
    }
    return (type == ORC_IQ3_S ? 1.0f : type == ORC_IQ3_XXS ? 0.25f : 0.125f) * hsum8(acc);
}
/* ---- IQ1_S / IQ1_M: per sub-block ib and 8-group l the grid index, the delta sign and the odd scale ---- */
#ifndef ORC_IQ1S_FMA
#define ORC_IQ1S_FMA 0        /* accum1 += d * sumi1: the reference build keeps the multiply and the add apart (1 fails the pin in tests/test_oracle_vs_reference.py) */
#endif
static void orc_iq1_vals8(uint32_t idx, int8_t v[8]) {
    const uint32_t code = IQ1S_CODE[idx], a = iq1_code_bytes4(code, 0), b = iq1_code_bytes4(code, 4);
    for (int k = 0; k < 4; k++) { v[k] = (int8_t)((a >> (8 * k)) & 0xff); v[4 + k] = (int8_t)((b >> (8 * k)) & 0xff); }
}
static uint32_t orc_iq1s_idx(const orc_block_iq1_s * b, int ib, int l) { return b->qs[4 * ib + l] | (((b->qh[ib] >> (3 * l)) & 7u) << 8); }
static uint32_t orc_iq1m_idx(const orc_block_iq1_m * b, int ib, int l) { const uint32_t h = b->qh[2 * ib + (l >> 1)]; return b->qs[4 * ib + l] | (((l & 1) ? (h << 4) : (h << 8)) & 0x700u); }
static int      orc_iq1m_neg(const orc_block_iq1_m * b, int ib, int l) { return (b->qh[2 * ib + (l >> 1)] & ((l & 1) ? 0x80 : 0x08)) != 0; }
static int      orc_iq1m_ls(const orc_block_iq1_m * b, int ib, int h) { uint16_t sc[4]; memcpy(sc, b->scales, 8); return 2 * ((sc[ib / 2] >> (6 * (ib % 2) + 3 * h)) & 7) + 1; }
static float    orc_iq1m_d(const orc_block_iq1_m * b) {
    uint16_t sc[4]; memcpy(sc, b->scales, 8);
    return orc_fp16_to_fp32((uint16_t)((sc[0] >> 12) | ((sc[1] >> 8) & 0x00f0) | ((sc[2] >> 4) & 0x0f00) | (sc[3] & 0xf000)));
}
void orc_dequantize_row_iq1_s(const orc_block_iq1_s * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int ib = 0; ib < 8; ib++) {
            const float dl = d * (float)(2 * ((x[i].qh[ib] >> 12) & 7) + 1), delta = (x[i].qh[ib] & 0x8000) ? -0.125f : 0.125f;
            for (int l = 0; l < 4; l++) {
                int8_t v[8]; orc_iq1_vals8(orc_iq1s_idx(&x[i], ib, l), v);
                for (int j = 0; j < 8; j++) y[i*256 + 32*ib + 8*l + j] = dl * ((float) v[j] + delta);
            }
        }
    }
}
void orc_dequantize_row_iq1_m(const orc_block_iq1_m * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_iq1m_d(&x[i]);
        for (int ib = 0; ib < 8; ib++)
            for (int l = 0; l < 4; l++) {
                const float dl = d * (float) orc_iq1m_ls(&x[i], ib, l >> 1), delta = orc_iq1m_neg(&x[i], ib, l) ? -0.125f : 0.125f;
                int8_t v[8]; orc_iq1_vals8(orc_iq1m_idx(&x[i], ib, l), v);
                for (int j = 0; j < 8; j++) y[i*256 + 32*ib + 8*l + j] = dl * ((float) v[j] + delta);
            }
    }
}
static inline float hsum8(const float x[8]);
float orc_vec_dot_iq1_s_q8_K_avx2(int64_t n, const orc_block_iq1_s * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0}, accum1 = 0.0f;
    for (int64_t i = 0; i < nb; i++) {
        int32_t sumi[8] = {0}, sumi1 = 0;
        for (int ib = 0; ib < 8; ib++) {
            const int ls = 2 * ((x[i].qh[ib] >> 12) & 7) + 1;
            for (int l = 0; l < 4; l++) {
                int8_t v[8]; orc_iq1_vals8(orc_iq1s_idx(&x[i], ib, l), v);
                for (int j = 0; j < 8; j++) sumi[2 * l + j / 4] += ls * (int) v[j] * (int) y[i].qs[32 * ib + 8 * l + j];
            }
            sumi1 += ((int) y[i].bsums[2 * ib] + (int) y[i].bsums[2 * ib + 1]) * ((x[i].qh[ib] & 0x8000) ? -1 : 1) * ls;
        }
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
        accum1 = ORC_IQ1S_FMA ? fmaf(d, (float) sumi1, accum1) : accum1 + d * (float) sumi1;
    }
    return hsum8(acc) + 0.125f * accum1;
}
float orc_vec_dot_iq1_m_q8_K_avx2(int64_t n, const orc_block_iq1_m * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc1[8] = {0}, acc2[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        int32_t s1[8] = {0}, s2[8] = {0};
        for (int ib = 0; ib < 8; ib++)
            for (int l = 0; l < 4; l++) {
                const int ls = orc_iq1m_ls(&x[i], ib, l >> 1), sg = orc_iq1m_neg(&x[i], ib, l) ? -1 : 1;
                int8_t v[8]; orc_iq1_vals8(orc_iq1m_idx(&x[i], ib, l), v);
                for (int j = 0; j < 8; j++) {
                    s1[2 * l + j / 4] += ls * (int) v[j] * (int) y[i].qs[32 * ib + 8 * l + j];
                    s2[2 * l + j / 4] += ls * sg * (int) y[i].qs[32 * ib + 8 * l + j];
                }
            }
        const float d = y[i].d * orc_iq1m_d(&x[i]);
        for (int L = 0; L < 8; L++) { acc1[L] = fmaf(d, (float) s1[L], acc1[L]); acc2[L] = fmaf(d, (float) s2[L], acc2[L]); }
    }
    return hsum8(acc1) + 0.125f * hsum8(acc2);
}
/* the trit of element e (dequantize order, ggml-quants.c:2215-2252): 160 elements from qs[0..31] (plane n = e / 32: byte * 3^n mod 256, times 3, top two bits), 80 from
 * qs[32..47] (planes of 16), 16 from qh (4 planes of 4) */
static int orc_tq1_trit(const orc_block_tq1_0 * b, int e) {
    static const uint8_t pow3[6] = {1, 3, 9, 27, 81, 243};
    uint8_t q;
    if (e < 160)      q = (uint8_t)(b->qs[e % 32] * pow3[e / 32]);
    else if (e < 240) q = (uint8_t)(b->qs[32 + (e - 160) % 16] * pow3[(e - 160) / 16]);
    else              q = (uint8_t)(b->qh[(e - 240) % 4] * pow3[(e - 240) / 4]);
    return (int)(((uint16_t) q * 3) >> 8);
}
static int orc_tq2_q(const orc_block_tq2_0 * b, int e) { return (b->qs[32 * (e / 128) + e % 32] >> (2 * ((e % 128) / 32))) & 3; }
void orc_dequantize_row_tq1_0(const orc_block_tq1_0 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int e = 0; e < 256; e++) y[i*256 + e] = (float)(orc_tq1_trit(&x[i], e) - 1) * d;
    }
}
void orc_dequantize_row_tq2_0(const orc_block_tq2_0 * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d);
        for (int e = 0; e < 256; e++) y[i*256 + e] = (float)(orc_tq2_q(&x[i], e) - 1) * d;
    }
}
/* Q2_K / Q3_K element (n128 = which 128, j = 2-bit plane 0..3, h = which half of the 32 bytes, l = 0..15): index n128 * 128 + j * 32 + h * 16 + l */
void orc_dequantize_row_q2_K(const orc_block_q2_K * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d), min = orc_fp16_to_fp32(x[i].dmin);
        for (int n = 0; n < 2; n++) for (int j = 0; j < 4; j++) for (int h = 0; h < 2; h++) {
            const uint8_t sc = x[i].scales[8 * n + 2 * j + h];
            const float dl = d * (float)(sc & 0xF), ml = min * (float)(sc >> 4);
            for (int l = 0; l < 16; l++) *y++ = dl * (float)((x[i].qs[32 * n + 16 * h + l] >> (2 * j)) & 3) - ml;
        }
    }
}
/* the sixteen 6-bit scales of a Q3_K super-block, minus 32 (ggml-quants.c:1141-1150) */
static void orc_q3k_scales(const uint8_t * sc12, int8_t * out) {
    uint32_t aux[4]; memcpy(aux, sc12, 12);
    const uint32_t kmask1 = 0x03030303u, kmask2 = 0x0f0f0f0fu, tmp = aux[2];
    aux[2] = ((aux[0] >> 4) & kmask2) | (((tmp >> 4) & kmask1) << 4);
    aux[3] = ((aux[1] >> 4) & kmask2) | (((tmp >> 6) & kmask1) << 4);
    aux[0] = (aux[0] & kmask2) | (((tmp >> 0) & kmask1) << 4);
    aux[1] = (aux[1] & kmask2) | (((tmp >> 2) & kmask1) << 4);
    memcpy(out, aux, 16);
    for (int s = 0; s < 16; s++) out[s] = (int8_t)(out[s] - 32);
}
void orc_dequantize_row_q3_K(const orc_block_q3_K * x, float * y, int64_t k) {
    const int64_t nb = k / ORC_QK_K;
    for (int64_t i = 0; i < nb; i++) {
        const float d_all = orc_fp16_to_fp32(x[i].d);
        int8_t sc[16]; orc_q3k_scales(x[i].scales, sc);
        for (int n = 0; n < 2; n++) for (int j = 0; j < 4; j++) for (int h = 0; h < 2; h++) {
            const float dl = d_all * (float) sc[8 * n + 2 * j + h];
            for (int l = 0; l < 16; l++) {
                const int b = 16 * h + l;
                const int q = (int)((x[i].qs[32 * n + b] >> (2 * j)) & 3) - ((x[i].hmask[b] >> (4 * n + j)) & 1 ? 0 : 4);
                *y++ = dl * (float) q;
            }
        }
    }
}
void orc_dequantize_row(int type, const void * x, float * y, int64_t k) {
    switch (type) {
        case ORC_Q4_0: orc_dequantize_row_q4_0((const orc_block_q4_0 *) x, y, k); break;
        case ORC_Q8_0: orc_dequantize_row_q8_0((const orc_block_q8_0 *) x, y, k); break;
        case ORC_Q4_1: orc_dequantize_row_q4_1((const orc_block_q4_1 *) x, y, k); break;
        case ORC_Q4_K: orc_dequantize_row_q4_K((const orc_block_q4_K *) x, y, k); break;
        case ORC_Q5_K: orc_dequantize_row_q5_K((const orc_block_q5_K *) x, y, k); break;
        case ORC_Q6_K: orc_dequantize_row_q6_K((const orc_block_q6_K *) x, y, k); break;
        case ORC_Q5_0: orc_dequantize_row_q5_0((const orc_block_q5_0 *) x, y, k); break;
        case ORC_Q5_1: orc_dequantize_row_q5_1((const orc_block_q5_1 *) x, y, k); break;
        case ORC_IQ4_NL: orc_dequantize_row_iq4_nl((const orc_block_iq4_nl *) x, y, k); break;
        case ORC_IQ4_XS: orc_dequantize_row_iq4_xs((const orc_block_iq4_xs *) x, y, k); break;
        case ORC_TQ1_0: orc_dequantize_row_tq1_0((const orc_block_tq1_0 *) x, y, k); break;
        case ORC_TQ2_0: orc_dequantize_row_tq2_0((const orc_block_tq2_0 *) x, y, k); break;
        case ORC_IQ2_XXS: case ORC_IQ2_XS: case ORC_IQ2_S: case ORC_IQ3_XXS: case ORC_IQ3_S: orc_dequantize_row_iq_grid(type, x, y, k); break;
        case ORC_IQ1_S: orc_dequantize_row_iq1_s((const orc_block_iq1_s *) x, y, k); break;
        case ORC_IQ1_M: orc_dequantize_row_iq1_m((const orc_block_iq1_m *) x, y, k); break;
        case ORC_MXFP4: orc_dequantize_row_mxfp4((const orc_block_mxfp4 *) x, y, k); break;
        case ORC_Q2_K: orc_dequantize_row_q2_K((const orc_block_q2_K *) x, y, k); break;
        case ORC_Q3_K: orc_dequantize_row_q3_K((const orc_block_q3_K *) x, y, k); break;
        case ORC_F16:  for (int64_t i = 0; i < k; i++) y[i] = orc_fp16_to_fp32(((const uint16_t *) x)[i]); break;
        case ORC_F32:  memcpy(y, x, (size_t) k * 4); break;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* dot products (generic branches of ggml-cpu/quants.c)                                        */
/* ------------------------------------------------------------------------------------------ */
float orc_vec_dot_q4_0_q8_0(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y, int32_t * isums) {
    const int64_t nb = n / ORC_QK;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ib++) {
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; j++) {
            s0 += ((x[ib].qs[j] & 0x0F) - 8) * y[ib].qs[j];
            s1 += ((x[ib].qs[j] >>   4) - 8) * y[ib].qs[j + 16];
        }
        const int sumi = s0 + s1;
        if (isums) isums[ib] = sumi;
        sumf += (float) sumi * orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d);
    }
    return sumf;
}

float orc_vec_dot_q4_1_q8_1(int64_t n, const orc_block_q4_1 * x, const orc_block_q8_1 * y, int32_t * isums) {
    const int64_t nb = n / ORC_QK;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ib++) {
        int s0 = 0, s1 = 0;
        for (int j = 0; j < 16; j++) {
            s0 += (x[ib].qs[j] & 0x0F) * y[ib].qs[j];
            s1 += (x[ib].qs[j] >>   4) * y[ib].qs[j + 16];
        }
        const int sumi = s0 + s1;
        if (isums) isums[ib] = sumi;
        sumf += (orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d)) * (float) sumi + orc_fp16_to_fp32(x[ib].m) * orc_fp16_to_fp32(y[ib].s);
    }
    return sumf;
}

float orc_vec_dot_q8_0_q8_0(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y, int32_t * isums) {
    const int64_t nb = n / ORC_QK;
    float sumf = 0;
    for (int64_t ib = 0; ib < nb; ib++) {
        int sumi = 0;
        for (int j = 0; j < 32; j++) sumi += x[ib].qs[j] * y[ib].qs[j];
        if (isums) isums[ib] = sumi;
        sumf += (float) sumi * (orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d));
    }
    return sumf;
}

/* quants.c:550-623.  The reference keeps 8 float lane-accumulators (sums[l] += d*aux32[l]) and
 * folds the mins term into sumf; we restate exactly that lane structure. */
float orc_vec_dot_q4_K_q8_K(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y, int32_t * isums) {
    const int64_t nb = n / ORC_QK_K;
    float sums[8] = {0};
    float sumf = 0;
    for (int64_t i = 0; i < nb; i++) {
        int8_t  a[ORC_QK_K];
        int32_t lane[8] = {0};
        const uint8_t * q4 = x[i].qs;
        for (int g = 0; g < 4; g++, q4 += 32) {
            for (int l = 0; l < 32; l++) a[g*64 + l]      = (int8_t)(q4[l] & 0xF);
            for (int l = 0; l < 32; l++) a[g*64 + 32 + l] = (int8_t)(q4[l] >> 4);
        }
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; j++) orc_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);

        int summ = 0;                                  /* sum_j bsums[j] * mins[j/2] */
        for (int j = 0; j < 16; j++) summ += y[i].bsums[j] * mn[j/2];

        const int8_t * q8 = y[i].qs;
        for (int j = 0; j < 8; j++) {                  /* 8 sub-blocks of 32 */
            for (int l = 0; l < 32; l++) lane[l & 7] += (int32_t) sc[j] * ((int16_t) q8[j*32 + l] * a[j*32 + l]);
        }
        if (isums) {
            int32_t t = 0;
            for (int l = 0; l < 8; l++) t += lane[l];
            isums[2*i + 0] = t;
            isums[2*i + 1] = summ;
        }
        const float d = orc_fp16_to_fp32(x[i].d) * y[i].d;
        for (int l = 0; l < 8; l++) sums[l] += d * (float) lane[l];
        const float dmin = orc_fp16_to_fp32(x[i].dmin) * y[i].d;
        sumf -= dmin * (float) summ;
    }
    for (int l = 0; l < 8; l++) sumf += sums[l];
    return sumf;
}

/* ------------------------------------------------------------------------------------------ */
/* dot products, x86 AVX2 branches (ggml-cpu/arch/x86/quants.c) -- LANE-EXACT restatements      */
/*                                                                                            */
/* The reference build on the benchmark host takes the AVX2 branches: 8 fp32 lane accumulators */
/* (lane L = dword L of every 32-byte chunk), ONE fused multiply-add per block and lane in      */
/* block order, then hsum_float_8.  Integer sums are exact in any order; the fp32 chain is      */
/* not, so these functions restate its order -- they are bit-identical to libggml-cpu.so       */
/* (tests/test_oracle_vs_reference.py), for every N: tinyBLAS_Q0_AVX (llamafile/sgemm.cpp:      */
/* 1346-1790, the n >= 2 path of Q4_0 / Q8_0) keeps the same per-element chain.                */
/* ------------------------------------------------------------------------------------------ */
static int g_order = ORC_ORDER_AVX2;
void orc_set_order(int order) { g_order = order; }
int  orc_get_order(void) { return g_order; }

/* hsum_float_8 (arch/x86/quants.c:43-49) */
static inline float hsum8(const float x[8]) {
    float r0 = x[4] + x[0], r1 = x[5] + x[1], r2 = x[6] + x[2], r3 = x[7] + x[3];
    r0 = r0 + r2; r1 = r1 + r3;
    return r0 + r1;
}

/* arch/x86/quants.c:543-577: lanes 0..3 = elements 0..15 (low nibbles), lanes 4..7 = elements 16..31 (high nibbles) */
float orc_vec_dot_q4_0_q8_0_avx2(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y) {
    const int64_t nb = n / ORC_QK;
    float acc[8] = {0};
    for (int64_t ib = 0; ib < nb; ib++) {
        const float d = orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d);
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) {
                const int i = 4 * L + e;
                const int w = i < 16 ? (x[ib].qs[i] & 0x0F) - 8 : (x[ib].qs[i - 16] >> 4) - 8;
                s += w * y[ib].qs[i];
            }
            acc[L] = fmaf(d, (float) s, acc[L]);
        }
    }
    return hsum8(acc);
}

/* arch/x86/quants.c:1012-1040 */
float orc_vec_dot_q8_0_q8_0_avx2(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y) {
    const int64_t nb = n / ORC_QK;
    float acc[8] = {0};
    for (int64_t ib = 0; ib < nb; ib++) {
        const float d = orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d);
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) s += x[ib].qs[4 * L + e] * y[ib].qs[4 * L + e];
            acc[L] = fmaf(d, (float) s, acc[L]);
        }
    }
    return hsum8(acc);
}

/* arch/x86/quants.c:701-760: summs += m_w * s_a is a scalar statement that gcc -O3 contracts into one fma
 * (-ffp-contract=fast is gcc's default for GNU C; verified bit for bit against the reference build) */
float orc_vec_dot_q4_1_q8_1_avx2(int64_t n, const orc_block_q4_1 * x, const orc_block_q8_1 * y) {
    const int64_t nb = n / ORC_QK;
    float acc[8] = {0};
    float summs = 0.0f;
    for (int64_t ib = 0; ib < nb; ib++) {
        const float d0 = orc_fp16_to_fp32(x[ib].d), d1 = orc_fp16_to_fp32(y[ib].d);
        summs = ORC_Q41_SUMMS(orc_fp16_to_fp32(x[ib].m), orc_fp16_to_fp32(y[ib].s), summs);
        const float d0d1 = d0 * d1;
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) {
                const int i = 4 * L + e;
                const int w = i < 16 ? (x[ib].qs[i] & 0x0F) : (x[ib].qs[i - 16] >> 4);
                s += w * y[ib].qs[i];
            }
            acc[L] = fmaf(d0d1, (float) s, acc[L]);
        }
    }
    return hsum8(acc) + summs;
}

/* arch/x86/quants.c:1742-1822: per super-block and lane L, sumi[L] = sum over the four 64-weight chunks of
 * sc_lo * (q4l . q8l)[dword L] + sc_hi * (q4h . q8h)[dword L]; acc[L] = fma(d, sumi[L], acc[L]);
 * acc_m[k] = fma(dmin, m[2k] S[2k] + m[2k+1] S[2k+1], acc_m[k]), S = bsums of the 32-weight sub-blocks, dmin = -(y.d) * x.dmin */
float orc_vec_dot_q4_K_q8_K_avx2(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0}, acc_m[4] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * orc_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; j++) orc_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        for (int k = 0; k < 4; k++) {
            const int S0 = y[i].bsums[4*k + 0] + y[i].bsums[4*k + 1], S1 = y[i].bsums[4*k + 2] + y[i].bsums[4*k + 3];
            const int prod = mn[2*k] * S0 + mn[2*k + 1] * S1;
            acc_m[k] = fmaf(dmin, (float) prod, acc_m[k]);
        }
        int32_t sumi[8] = {0};
        for (int c = 0; c < 4; c++) {
            const uint8_t * q4 = x[i].qs + 32 * c;
            const int8_t  * q8 = y[i].qs + 64 * c;
            for (int L = 0; L < 8; L++) {
                int pl = 0, ph = 0;
                for (int e = 0; e < 4; e++) {
                    pl += (q4[4*L + e] & 0x0F) * q8[4*L + e];
                    ph += (q4[4*L + e] >> 4)   * q8[32 + 4*L + e];
                }
                sumi[L] += sc[2*c] * pl + sc[2*c + 1] * ph;
            }
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    const float m02 = acc_m[0] + acc_m[2], m13 = acc_m[1] + acc_m[3];
    return hsum8(acc) + (m02 + m13);
}

#ifndef ORC_Q5K_SUMMS
#define ORC_Q5K_SUMMS(dmin, v, acc) ((acc) + (dmin) * (v))          /* `summs += dmin * hsum` as compiled in the reference build */
#endif
/* ggml_vec_dot_q5_K_q8_K, AVX2 (arch/x86/quants.c:1916-2030): Q4_K's lanes with a fifth bit from qh; the mins in ONE scalar chain */
float orc_vec_dot_q5_K_q8_K_avx2(int64_t n, const orc_block_q5_K * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0}, summs = 0.0f;
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * orc_fp16_to_fp32(x[i].dmin);
        uint8_t sc[8], mn[8];
        for (int j = 0; j < 8; j++) orc_scale_min_k4(j, x[i].scales, &sc[j], &mn[j]);
        int hsum = 0;
        for (int j = 0; j < 8; j++) hsum += mn[j] * (y[i].bsums[2*j] + y[i].bsums[2*j + 1]);
        summs = ORC_Q5K_SUMMS(dmin, (float) hsum, summs);
        int32_t sumi[8] = {0};
        for (int c = 0; c < 4; c++) {
            const uint8_t * q5 = x[i].qs + 32 * c, * qh = x[i].qh;
            const int8_t  * q8 = y[i].qs + 64 * c;
            for (int L = 0; L < 8; L++) {
                int pl = 0, ph = 0;
                for (int e = 0; e < 4; e++) {
                    const int b = 4*L + e;
                    pl += ((q5[b] & 0x0F) + (((qh[b] >> (2*c))     & 1) << 4)) * q8[b];
                    ph += ((q5[b] >> 4)   + (((qh[b] >> (2*c + 1)) & 1) << 4)) * q8[32 + b];
                }
                sumi[L] += sc[2*c] * pl + sc[2*c + 1] * ph;
            }
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    return hsum8(acc) + summs;
}
/* ggml_vec_dot_q6_K_q8_K, AVX2 (arch/x86/quants.c:2130-2225): per 128 elements four 32-element groups g, lane A takes bytes 4A..4A+3 of
 * each; the int8 scale of a group's first / second 16 elements goes to lanes 0..3 / 4..7 (get_scale_shuffle) */
float orc_vec_dot_q6_K_q8_K_avx2(int64_t n, const orc_block_q6_K * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        int32_t sumi[8] = {0};
        for (int j = 0; j < 2; j++) {
            const uint8_t * ql = x[i].ql + 64 * j, * qh = x[i].qh + 32 * j;
            const int8_t * q8 = y[i].qs + 128 * j, * sc = x[i].scales + 8 * j;
            for (int g = 0; g < 4; g++)
                for (int L = 0; L < 8; L++) {
                    int p = 0;
                    for (int e = 0; e < 4; e++) {
                        const int b = 4*L + e;
                        const int lo = g < 2 ? (ql[32 * (g & 1) + b] & 0xF) : (ql[32 * (g & 1) + b] >> 4);
                        const int q = (lo | (((qh[b] >> (2*g)) & 3) << 4)) - 32;
                        p += q * q8[32 * g + b];
                    }
                    sumi[L] += sc[2*g + (L >> 2)] * p;
                }
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    return hsum8(acc);
}


/* ggml_vec_dot_iq4_xs_q8_K, AVX2 (arch/x86/quants.c:3716-3764): sub-block ib's 32 codebook values (elements 0..15 = low nibbles, 16..31 = high nibbles of its 16 bytes)
 * against the activation's 32 int8: lane A sums elements 4A..4A+3 (mul_add_epi8 + madd_epi16 with the sub-block's scale ls - 32: exact integers), sumi1 / sumi2 (even /
 * odd sub-blocks) are added as integers, then one fma per super-block and lane */
float orc_vec_dot_iq4_xs_q8_K_avx2(int64_t n, const orc_block_iq4_xs * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = orc_fp16_to_fp32(x[i].d) * y[i].d;
        int32_t sumi[8] = {0};
        for (int ib = 0; ib < 8; ib++) {
            const int ls = (((x[i].scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((x[i].scales_h >> (2 * ib)) & 3) << 4)) - 32;
            for (int L = 0; L < 8; L++) {
                int p = 0;
                for (int e = 0; e < 4; e++) p += (int) orc_kvalues_iq4nl[orc_nib_elem(x[i].qs + 16 * ib, 4 * L + e)] * (int) y[i].qs[32 * ib + 4 * L + e];
                sumi[L] += ls * p;
            }
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    return hsum8(acc);
}

/* ggml_vec_dot_tq1_0_q8_K / ggml_vec_dot_tq2_0_q8_K, AVX2 (arch/x86/quants.c:1080-1210, 1212-1270): 16-bit lane sums that cannot overflow, bsums subtracted LANE-wise
 * (16-bit lane t takes bsums[t]), madd with ones (lanes 2L, 2L + 1 -> L), then (float) sumi * d + sumf: a multiply and an add in the source, and the reference build keeps them apart (two roundings; ORC_TQ_FMA = 1 fails the pin) */
#ifndef ORC_TQ_FMA
#define ORC_TQ_FMA 0
#endif
static float orc_tq_fold(float d, int32_t sumi, float acc) { return ORC_TQ_FMA ? fmaf((float) sumi, d, acc) : (float) sumi * d + acc; }
float orc_vec_dot_tq1_0_q8_K_avx2(int64_t n, const orc_block_tq1_0 * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        for (int L = 0; L < 8; L++) {
            int32_t s = 0;
            for (int c = 0; c < 8; c++)
                for (int e = 0; e < 4; e++) s += orc_tq1_trit(&x[i], 32 * c + 4 * L + e) * (int) y[i].qs[32 * c + 4 * L + e];
            s -= (int) y[i].bsums[2 * L] + (int) y[i].bsums[2 * L + 1];
            acc[L] = orc_tq_fold(d, s, acc[L]);
        }
    }
    return hsum8(acc);
}
float orc_vec_dot_tq2_0_q8_K_avx2(int64_t n, const orc_block_tq2_0 * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        for (int L = 0; L < 8; L++) {
            int32_t s = 0;
            for (int c = 0; c < 8; c++)
                for (int e = 0; e < 4; e++) s += orc_tq2_q(&x[i], 32 * c + 4 * L + e) * (int) y[i].qs[32 * c + 4 * L + e];
            s -= (int) y[i].bsums[2 * L] + (int) y[i].bsums[2 * L + 1];
            acc[L] = orc_tq_fold(d, s, acc[L]);
        }
    }
    return hsum8(acc);
}

/* arch/x86/quants.c:846-884: Q4_0's lanes and chain with 5-bit values (nib | bit << 4) - 16 */
float orc_vec_dot_q5_0_q8_0_avx2(int64_t n, const orc_block_q5_0 * x, const orc_block_q8_0 * y) {
    const int64_t nb = n / ORC_QK;
    float acc[8] = {0};
    for (int64_t ib = 0; ib < nb; ib++) {
        const float d = orc_fp16_to_fp32(x[ib].d) * orc_fp16_to_fp32(y[ib].d);
        const uint32_t qh = orc_qh32(x[ib].qh);
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) s += (orc_q5_elem(x[ib].qs, qh, 4 * L + e) - 16) * y[ib].qs[4 * L + e];
            acc[L] = fmaf(d, (float) s, acc[L]);
        }
    }
    return hsum8(acc);
}
/* arch/x86/quants.c:926-968: Q4_1's lanes and chains (the scalar `summs += m * s` contracted by gcc, as for Q4_1) */
float orc_vec_dot_q5_1_q8_1_avx2(int64_t n, const orc_block_q5_1 * x, const orc_block_q8_1 * y) {
    const int64_t nb = n / ORC_QK;
    float acc[8] = {0};
    float summs = 0.0f;
    for (int64_t ib = 0; ib < nb; ib++) {
        const float dx = orc_fp16_to_fp32(x[ib].d), dy = orc_fp16_to_fp32(y[ib].d);
        summs = ORC_Q41_SUMMS(orc_fp16_to_fp32(x[ib].m), orc_fp16_to_fp32(y[ib].s), summs);
        const float dd = dx * dy;
        const uint32_t qh = orc_qh32(x[ib].qh);
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) s += orc_q5_elem(x[ib].qs, qh, 4 * L + e) * y[ib].qs[4 * L + e];
            acc[L] = fmaf((float) s, dd, acc[L]);
        }
    }
    return hsum8(acc) + summs;
}
/* the 16-entry codebook formats: arch/x86/quants.c:3632-3714 (IQ4_NL), 760-844 (MXFP4).  Blocks in pairs: even blocks accumulate in accum1, odd ones in
 * accum2; hsum_float_8(accum1 + accum2); a last unpaired block in scalar code: sumf += d * (sumi1 + sumi2) (contracted by gcc).  chain = 1: the single
 * accumulator of tinyBLAS_Q0_AVX (llamafile/sgemm.cpp:1346-1790), what mul_mat takes for IQ4_NL with >= 2 activation columns */
static float codebook_dot(int64_t nb, const uint8_t * xb, size_t xs, const int8_t * tab, int mx, const orc_block_q8_0 * y, int chain) {
    float acc1[8] = {0}, acc2[8] = {0};
    int64_t ib = 0;
    const int64_t npair = chain ? nb : (nb & ~(int64_t) 1);
    for (; ib < npair; ib++) {
        const uint8_t * b = xb + (size_t) ib * xs;
        const uint8_t * qs = mx ? b + 1 : b + 2;
        uint16_t dh; memcpy(&dh, b, 2);
        const float d = mx ? orc_fp16_to_fp32(y[ib].d) * orc_e8m0_half(b[0]) : orc_fp16_to_fp32(y[ib].d) * orc_fp16_to_fp32(dh);
        float * acc = (chain || !(ib & 1)) ? acc1 : acc2;
        for (int L = 0; L < 8; L++) {
            int s = 0;
            for (int e = 0; e < 4; e++) s += tab[orc_nib_elem(qs, 4 * L + e)] * y[ib].qs[4 * L + e];
            acc[L] = fmaf(d, (float) s, acc[L]);
        }
    }
    float sum[8];
    for (int L = 0; L < 8; L++) sum[L] = chain ? acc1[L] : acc1[L] + acc2[L];
    float sumf = hsum8(sum);
    for (; ib < nb; ib++) {
        const uint8_t * b = xb + (size_t) ib * xs;
        const uint8_t * qs = mx ? b + 1 : b + 2;
        uint16_t dh; memcpy(&dh, b, 2);
        const float d = mx ? orc_fp16_to_fp32(y[ib].d) * orc_e8m0_half(b[0]) : orc_fp16_to_fp32(y[ib].d) * orc_fp16_to_fp32(dh);
        int s = 0;
        for (int e = 0; e < 32; e++) s += tab[orc_nib_elem(qs, e)] * y[ib].qs[e];
        sumf = ORC_CODEBOOK_TAIL(d, (float) s, sumf);
    }
    return sumf;
}
float orc_vec_dot_iq4_nl_q8_0_avx2(int64_t n, const orc_block_iq4_nl * x, const orc_block_q8_0 * y, int chain) {
    return codebook_dot(n / ORC_QK, (const uint8_t *) x, sizeof(orc_block_iq4_nl), orc_kvalues_iq4nl, 0, y, chain);
}
float orc_vec_dot_mxfp4_q8_0_avx2(int64_t n, const orc_block_mxfp4 * x, const orc_block_q8_0 * y) {
    return codebook_dot(n / ORC_QK, (const uint8_t *) x, sizeof(orc_block_mxfp4), orc_kvalues_mxfp4, 1, y, 0);
}
/* arch/x86/quants.c:1278-1354.  Per 128 weights one 32-byte vector of 2-bit fields; plane j (bits 2j) times the activations 32 j .. 32 j + 31 of the 128; lane A
 * = bytes 4A..4A+3 of the vector, its scale the one of the 16-weight sub-block it lies in: scales[8 n + 2 j + (A >> 2)] */
float orc_vec_dot_q2_K_q8_K_avx2(int64_t n, const orc_block_q2_K * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        const float dmin = -y[i].d * orc_fp16_to_fp32(x[i].dmin);
        for (int L = 0; L < 8; L++) {
            const int prod = (x[i].scales[2*L] >> 4) * y[i].bsums[2*L] + (x[i].scales[2*L + 1] >> 4) * y[i].bsums[2*L + 1];
            acc[L] = fmaf(dmin, (float) prod, acc[L]);
        }
        int32_t sumi[8] = {0};
        for (int nn = 0; nn < 2; nn++) for (int j = 0; j < 4; j++) for (int L = 0; L < 8; L++) {
            int p = 0;
            for (int e = 0; e < 4; e++) {
                const int b = 4*L + e;
                p += (int)((x[i].qs[32 * nn + b] >> (2 * j)) & 3) * y[i].qs[128 * nn + 32 * j + b];
            }
            sumi[L] += (x[i].scales[8 * nn + 2 * j + (L >> 2)] & 0xF) * p;
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    return hsum8(acc);
}
/* arch/x86/quants.c:1470-1580: the same lanes; value = 2-bit field - (hmask bit clear ? 4 : 0), scale = 6-bit - 32 */
float orc_vec_dot_q3_K_q8_K_avx2(int64_t n, const orc_block_q3_K * x, const orc_block_q8_K * y) {
    const int64_t nb = n / ORC_QK_K;
    float acc[8] = {0};
    for (int64_t i = 0; i < nb; i++) {
        const float d = y[i].d * orc_fp16_to_fp32(x[i].d);
        int8_t sc[16]; orc_q3k_scales(x[i].scales, sc);
        int32_t sumi[8] = {0};
        for (int nn = 0; nn < 2; nn++) for (int j = 0; j < 4; j++) for (int L = 0; L < 8; L++) {
            int p = 0;
            for (int e = 0; e < 4; e++) {
                const int b = 4*L + e;
                const int q = (int)((x[i].qs[32 * nn + b] >> (2 * j)) & 3) - ((x[i].hmask[b] >> (4 * nn + j)) & 1 ? 0 : 4);
                p += q * y[i].qs[128 * nn + 32 * j + b];
            }
            sumi[L] += sc[8 * nn + 2 * j + (L >> 2)] * p;
        }
        for (int L = 0; L < 8; L++) acc[L] = fmaf(d, (float) sumi[L], acc[L]);
    }
    return hsum8(acc);
}

/* ggml_vec_dot_f16, AVX2 + F16C (ggml-cpu/vec.cpp:264-, simd-mappings.h:528-620): four 8-lane accumulators over steps of 32,
 * GGML_F32x8_REDUCE, leftovers in double */
float orc_vec_dot_f16_avx2(int64_t n, const uint16_t * x, const uint16_t * y) {
    const int64_t np = n & ~(int64_t) 31;
    float sum[4][8] = {{0}};
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++) sum[j][l] = fmaf(orc_fp16_to_fp32(x[i + 8*j + l]), orc_fp16_to_fp32(y[i + 8*j + l]), sum[j][l]);
    float t0[4];
    for (int l = 0; l < 8; l++) { sum[0][l] = sum[0][l] + sum[2][l]; sum[1][l] = sum[1][l] + sum[3][l]; }
    for (int l = 0; l < 8; l++) sum[0][l] = sum[0][l] + sum[1][l];
    for (int l = 0; l < 4; l++) t0[l] = sum[0][l] + sum[0][l + 4];
    double sumf = (double)((t0[0] + t0[1]) + (t0[2] + t0[3]));
    for (int64_t i = np; i < n; i++) sumf += (double)(orc_fp16_to_fp32(x[i]) * orc_fp16_to_fp32(y[i]));
    return (float) sumf;
}
/* ggml_vec_dot_f32, AVX2 (ggml-cpu/vec.cpp:11-): the same lanes; leftovers `sumf += x[i]*y[i]` in float (contracted by gcc) */
float orc_vec_dot_f32_avx2(int64_t n, const float * x, const float * y) {
    const int64_t np = n & ~(int64_t) 31;
    float sum[4][8] = {{0}};
    for (int64_t i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++) sum[j][l] = fmaf(x[i + 8*j + l], y[i + 8*j + l], sum[j][l]);
    float t0[4];
    for (int l = 0; l < 8; l++) { sum[0][l] = sum[0][l] + sum[2][l]; sum[1][l] = sum[1][l] + sum[3][l]; }
    for (int l = 0; l < 8; l++) sum[0][l] = sum[0][l] + sum[1][l];
    for (int l = 0; l < 4; l++) t0[l] = sum[0][l] + sum[0][l + 4];
    float sumf = (t0[0] + t0[1]) + (t0[2] + t0[3]);
    /* the leftover loop `sumf += x[i]*y[i]` as gcc -O3 compiles it: groups of ORC_F32_TAIL_VW elements have their products formed by a vector
     * multiply (rounded) and added in order; the scalar remainder is contracted into fmas */
    int64_t i = np;
    for (; i + ORC_F32_TAIL_VW <= n; i += ORC_F32_TAIL_VW) for (int l = 0; l < ORC_F32_TAIL_VW; l++) sumf = sumf + x[i + l] * y[i + l];
    for (; i < n; i++) sumf = fmaf(x[i], y[i], sumf);
    return sumf;
}

/* tinyBLAS<8, __m256, ...> (llamafile/sgemm.cpp:477-640): the n >= 2 path of F16 / F32 src0 when k % 8 == 0 and m % 4 == 0:
 * ONE 8-lane accumulator per output element over steps of 8, hsum (:247-266 == hsum_float_8) */
static float tiny8_f16(int64_t n, const uint16_t * x, const uint16_t * y) {
    float acc[8] = {0};
    for (int64_t i = 0; i < n; i += 8) for (int l = 0; l < 8; l++) acc[l] = fmaf(orc_fp16_to_fp32(x[i + l]), orc_fp16_to_fp32(y[i + l]), acc[l]);
    return hsum8(acc);
}
static float tiny8_f32(int64_t n, const float * x, const float * y) {
    float acc[8] = {0};
    for (int64_t i = 0; i < n; i += 8) for (int l = 0; l < 8; l++) acc[l] = fmaf(x[i + l], y[i + l], acc[l]);
    return hsum8(acc);
}

/* ------------------------------------------------------------------------------------------ */
/* helpers for strided tensors                                                                 */
/* ------------------------------------------------------------------------------------------ */
static inline char * tptr(const orc_tensor * t, int64_t i0, int64_t i1, int64_t i2, int64_t i3) {
    return (char *) t->data + i0*(int64_t)t->nb[0] + i1*(int64_t)t->nb[1] + i2*(int64_t)t->nb[2] + i3*(int64_t)t->nb[3];
}
static inline int64_t nrows(const orc_tensor * t) { return t->ne[1]*t->ne[2]*t->ne[3]; }
static int is_contiguous(const orc_tensor * t) {
    size_t nb = orc_type_size(t->type);
    if (t->nb[0] != nb) return 0;
    nb = nb * (size_t)(t->ne[0] / orc_blck_size(t->type));
    for (int i = 1; i < 4; i++) { if (t->ne[i] != 1 && t->nb[i] != nb) return 0; nb *= (size_t) t->ne[i]; }
    return 1;
}


/* ------------------------------------------------------------------------------------------ */
/* weight quantizers: the reference's from_float_ref (ggml-quants.c:36-197, 622-702, 1280-1350), what chatllm.cpp's loader calls when a    */
/* tensor is re-quantized on load (src/chat.cpp:1246-1279 -> ggml::from_float, src/layers.cpp:358-373).  Plain C in the reference too       */
/* (ISO C mode: no contraction), so every operation below is one rounding in the written order.                                             */
/* ------------------------------------------------------------------------------------------ */
/* the element of largest magnitude (the FIRST one on ties) decides sign and scale: d = max / -(2^(bits-1)) */
static float signed_absmax(const float * x, int n) {
    float amax = 0.0f, max = 0.0f;
    for (int j = 0; j < n; j++) { const float v = x[j]; if (amax < fabsf(v)) { amax = fabsf(v); max = v; } }
    return max;
}
static void minmax(const float * x, int n, float * mn, float * mx) {
    float a = 3.402823466e+38f, b = -3.402823466e+38f;
    for (int j = 0; j < n; j++) { const float v = x[j]; if (v < a) a = v; if (v > b) b = v; }
    *mn = a; *mx = b;
}
void orc_quantize_row_q4_0_ref(const float * x, orc_block_q4_0 * y, int64_t k) {      /* ggml-quants.c:36-71 */
    for (int64_t i = 0; i < k / ORC_QK; i++, x += ORC_QK) {
        const float d = signed_absmax(x, ORC_QK) / -8, id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        for (int j = 0; j < 16; j++) {
            const float x0 = x[j] * id, x1 = x[j + 16] * id;
            const uint8_t q0 = (uint8_t) ORC_MIN(15, (int8_t)(x0 + 8.5f)), q1 = (uint8_t) ORC_MIN(15, (int8_t)(x1 + 8.5f));
            y[i].qs[j] = (uint8_t)(q0 | (q1 << 4));
        }
    }
}
void orc_quantize_row_q4_1_ref(const float * x, orc_block_q4_1 * y, int64_t k) {      /* ggml-quants.c:73-108 */
    for (int64_t i = 0; i < k / ORC_QK; i++, x += ORC_QK) {
        float mn, mx; minmax(x, ORC_QK, &mn, &mx);
        const float d = (mx - mn) / 15, id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d); y[i].m = orc_fp32_to_fp16(mn);
        for (int j = 0; j < 16; j++) {
            const float x0 = (x[j] - mn) * id, x1 = (x[j + 16] - mn) * id;
            const uint8_t q0 = (uint8_t) ORC_MIN(15, (int8_t)(x0 + 0.5f)), q1 = (uint8_t) ORC_MIN(15, (int8_t)(x1 + 0.5f));
            y[i].qs[j] = (uint8_t)(q0 | (q1 << 4));
        }
    }
}
void orc_quantize_row_q5_0_ref(const float * x, orc_block_q5_0 * y, int64_t k) {      /* ggml-quants.c:110-152 */
    for (int64_t i = 0; i < k / ORC_QK; i++, x += ORC_QK) {
        const float d = signed_absmax(x, ORC_QK) / -16, id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d);
        uint32_t qh = 0;
        for (int j = 0; j < 16; j++) {
            const float x0 = x[j] * id, x1 = x[j + 16] * id;
            const uint8_t q0 = (uint8_t) ORC_MIN(31, (int8_t)(x0 + 16.5f)), q1 = (uint8_t) ORC_MIN(31, (int8_t)(x1 + 16.5f));
            y[i].qs[j] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
            qh |= (uint32_t)((q0 & 0x10u) >> 4) << j;
            qh |= (uint32_t)((q1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}
void orc_quantize_row_q5_1_ref(const float * x, orc_block_q5_1 * y, int64_t k) {      /* ggml-quants.c:154-197 */
    for (int64_t i = 0; i < k / ORC_QK; i++, x += ORC_QK) {
        float mn, mx; minmax(x, ORC_QK, &mn, &mx);
        const float d = (mx - mn) / 31, id = d ? 1.0f / d : 0.0f;
        y[i].d = orc_fp32_to_fp16(d); y[i].m = orc_fp32_to_fp16(mn);
        uint32_t qh = 0;
        for (int j = 0; j < 16; j++) {
            const float x0 = (x[j] - mn) * id, x1 = (x[j + 16] - mn) * id;
            const uint8_t q0 = (uint8_t)(x0 + 0.5f), q1 = (uint8_t)(x1 + 0.5f);
            y[i].qs[j] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
            qh |= (uint32_t)((q0 & 0x10u) >> 4) << j;
            qh |= (uint32_t)((q1 & 0x10u) >> 4) << (j + 16);
        }
        memcpy(y[i].qh, &qh, 4);
    }
}
/* make_qkx2_quants (ggml-quants.c:622-702) with the arguments quantize_row_q4_K_ref passes: n = 32, nmax = 15, rmin = -1, rdelta = 0.1, nstep = 20, squared error.
 * Weighted least squares for x ~ scale * L + min over L in 0..15: a first guess from the range, then 21 candidate scalings of the rounding grid; returns scale, *the_min = -min */
static float qkx2_32(const float * x, const float * w, uint8_t * L, float * the_min) {
    uint8_t Laux[32];
    float min = x[0], max = x[0], sum_w = w[0], sum_x = sum_w * x[0];
    for (int i = 1; i < 32; i++) {
        if (x[i] < min) min = x[i];
        if (x[i] > max) max = x[i];
        sum_w += w[i];
        sum_x += w[i] * x[i];
    }
    if (min > 0) min = 0;
    if (max == min) { memset(L, 0, 32); *the_min = -min; return 0.0f; }
    float iscale = 15 / (max - min), scale = 1 / iscale, best_error = 0;
    for (int i = 0; i < 32; i++) {
        const int l = orc_nearest_int(iscale * (x[i] - min));
        L[i] = (uint8_t) ORC_MAX(0, ORC_MIN(15, l));
        float diff = scale * L[i] + min - x[i];
        diff = diff * diff;
        best_error += w[i] * diff;
    }
    for (int is = 0; is <= 20; is++) {
        iscale = (-1.0f + 0.1f * is + 15) / (max - min);
        float sum_l = 0, sum_l2 = 0, sum_xl = 0;
        for (int i = 0; i < 32; i++) {
            int l = orc_nearest_int(iscale * (x[i] - min));
            l = ORC_MAX(0, ORC_MIN(15, l));
            Laux[i] = (uint8_t) l;
            sum_l += w[i] * l;
            sum_l2 += w[i] * l * l;
            sum_xl += w[i] * l * x[i];
        }
        const float D = sum_w * sum_l2 - sum_l * sum_l;
        if (D > 0) {
            float this_scale = (sum_w * sum_xl - sum_x * sum_l) / D, this_min = (sum_l2 * sum_x - sum_l * sum_xl) / D;
            if (this_min > 0) { this_min = 0; this_scale = sum_xl / sum_l2; }
            float cur_error = 0;
            for (int i = 0; i < 32; i++) {
                float diff = this_scale * Laux[i] + this_min - x[i];
                diff = diff * diff;
                cur_error += w[i] * diff;
            }
            if (cur_error < best_error) { memcpy(L, Laux, 32); best_error = cur_error; scale = this_scale; min = this_min; }
        }
    }
    *the_min = -min;
    return scale;
}
void orc_quantize_row_q4_K_ref(const float * x, orc_block_q4_K * y, int64_t k) {      /* ggml-quants.c:1280-1350 */
    for (int64_t i = 0; i < k / ORC_QK_K; i++, x += ORC_QK_K) {
        uint8_t L[ORC_QK_K];
        float scales[8], mins[8], max_scale = 0, max_min = 0;
        for (int j = 0; j < 8; j++) {
            float w[32], sum_x2 = 0;
            for (int l = 0; l < 32; l++) sum_x2 += x[32*j + l] * x[32*j + l];
            const float av_x = sqrtf(sum_x2 / 32);
            for (int l = 0; l < 32; l++) w[l] = av_x + fabsf(x[32*j + l]);
            scales[j] = qkx2_32(x + 32*j, w, L + 32*j, &mins[j]);
            if (scales[j] > max_scale) max_scale = scales[j];
            if (mins[j] > max_min) max_min = mins[j];
        }
        const float inv_scale = max_scale > 0 ? 63.f / max_scale : 0.f, inv_min = max_min > 0 ? 63.f / max_min : 0.f;
        memset(y[i].scales, 0, 12);
        for (int j = 0; j < 8; j++) {
            uint8_t ls = (uint8_t) orc_nearest_int(inv_scale * scales[j]), lm = (uint8_t) orc_nearest_int(inv_min * mins[j]);
            ls = ORC_MIN(63, ls); lm = ORC_MIN(63, lm);
            if (j < 4) { y[i].scales[j] = ls; y[i].scales[j + 4] = lm; }
            else { y[i].scales[j + 4] = (uint8_t)((ls & 0xF) | ((lm & 0xF) << 4)); y[i].scales[j - 4] |= (uint8_t)((ls >> 4) << 6); y[i].scales[j] |= (uint8_t)((lm >> 4) << 6); }
        }
        y[i].d = orc_fp32_to_fp16(max_scale / 63.f); y[i].dmin = orc_fp32_to_fp16(max_min / 63.f);
        for (int j = 0; j < 8; j++) {
            uint8_t sc, m; orc_scale_min_k4(j, y[i].scales, &sc, &m);
            const float d = orc_fp16_to_fp32(y[i].d) * sc;
            if (!d) continue;                                                       /* (L keeps the search's values) */
            const float dm = orc_fp16_to_fp32(y[i].dmin) * m;
            for (int ii = 0; ii < 32; ii++) {
                const int l = orc_nearest_int((x[32*j + ii] + dm) / d);
                L[32*j + ii] = (uint8_t) ORC_MAX(0, ORC_MIN(15, l));
            }
        }
        for (int j = 0; j < 4; j++) for (int l = 0; l < 32; l++) y[i].qs[32*j + l] = (uint8_t)(L[64*j + l] | (L[64*j + 32 + l] << 4));
    }
}
int orc_quantize_row_ref(int type, const float * x, void * y, int64_t k) {
    switch (type) {
        case ORC_Q8_0: orc_quantize_row_q8_0_ref(x, (orc_block_q8_0 *) y, k); return 0;
        case ORC_Q4_0: orc_quantize_row_q4_0_ref(x, (orc_block_q4_0 *) y, k); return 0;
        case ORC_Q4_1: orc_quantize_row_q4_1_ref(x, (orc_block_q4_1 *) y, k); return 0;
        case ORC_Q5_0: orc_quantize_row_q5_0_ref(x, (orc_block_q5_0 *) y, k); return 0;
        case ORC_Q5_1: orc_quantize_row_q5_1_ref(x, (orc_block_q5_1 *) y, k); return 0;
        case ORC_Q4_K: orc_quantize_row_q4_K_ref(x, (orc_block_q4_K *) y, k); return 0;
        case ORC_F16:  for (int64_t i = 0; i < k; i++) ((uint16_t *) y)[i] = orc_fp32_to_fp16(x[i]); return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* mul_mat                                                                                     */
/* ------------------------------------------------------------------------------------------ */
static int vec_dot_type_of(int wtype) {
    switch (wtype) {             /* type_traits_cpu[], ggml-cpu/ggml-cpu.c:207-390 */
        case ORC_Q4_0: case ORC_Q8_0: case ORC_Q5_0: case ORC_IQ4_NL: case ORC_MXFP4: return ORC_Q8_0;
        case ORC_Q4_1: case ORC_Q5_1: return ORC_Q8_1;
        case ORC_Q4_K: case ORC_Q5_K: case ORC_Q6_K: case ORC_Q2_K: case ORC_Q3_K: case ORC_IQ4_XS: case ORC_TQ1_0: case ORC_TQ2_0: case ORC_IQ2_XXS: case ORC_IQ2_XS: case ORC_IQ2_S: case ORC_IQ3_XXS: case ORC_IQ3_S: case ORC_IQ1_S: case ORC_IQ1_M: return ORC_Q8_K;
        case ORC_F16:  return ORC_F16;
        case ORC_F32:  return ORC_F32;
    }
    return -1;
}

static void convert_row(int vtype, const float * x, void * y, int64_t k) {
    switch (vtype) {
        case ORC_Q8_0: orc_quantize_row_q8_0(x, (orc_block_q8_0 *) y, k); break;
        case ORC_Q8_1: orc_quantize_row_q8_1(x, (orc_block_q8_1 *) y, k); break;
        case ORC_Q8_K: orc_quantize_row_q8_K(x, (orc_block_q8_K *) y, k); break;
        case ORC_F16:  for (int64_t i = 0; i < k; i++) ((uint16_t *) y)[i] = orc_fp32_to_fp16(x[i]); break;
        case ORC_F32:  memcpy(y, x, (size_t) k * 4); break;
    }
}

static float vec_dot(int wtype, int64_t n, const void * w, const void * a) {
    switch (wtype) {
        case ORC_Q4_0: return g_order == ORC_ORDER_AVX2 ? orc_vec_dot_q4_0_q8_0_avx2(n, (const orc_block_q4_0 *) w, (const orc_block_q8_0 *) a)
                                                        : orc_vec_dot_q4_0_q8_0(n, (const orc_block_q4_0 *) w, (const orc_block_q8_0 *) a, NULL);
        case ORC_Q8_0: return g_order == ORC_ORDER_AVX2 ? orc_vec_dot_q8_0_q8_0_avx2(n, (const orc_block_q8_0 *) w, (const orc_block_q8_0 *) a)
                                                        : orc_vec_dot_q8_0_q8_0(n, (const orc_block_q8_0 *) w, (const orc_block_q8_0 *) a, NULL);
        case ORC_Q4_1: return g_order == ORC_ORDER_AVX2 ? orc_vec_dot_q4_1_q8_1_avx2(n, (const orc_block_q4_1 *) w, (const orc_block_q8_1 *) a)
                                                        : orc_vec_dot_q4_1_q8_1(n, (const orc_block_q4_1 *) w, (const orc_block_q8_1 *) a, NULL);
        case ORC_Q4_K: return g_order == ORC_ORDER_AVX2 ? orc_vec_dot_q4_K_q8_K_avx2(n, (const orc_block_q4_K *) w, (const orc_block_q8_K *) a)
                                                        : orc_vec_dot_q4_K_q8_K(n, (const orc_block_q4_K *) w, (const orc_block_q8_K *) a, NULL);
        case ORC_Q5_K: return orc_vec_dot_q5_K_q8_K_avx2(n, (const orc_block_q5_K *) w, (const orc_block_q8_K *) a);
        case ORC_Q6_K: return orc_vec_dot_q6_K_q8_K_avx2(n, (const orc_block_q6_K *) w, (const orc_block_q8_K *) a);
        case ORC_Q5_0: return orc_vec_dot_q5_0_q8_0_avx2(n, (const orc_block_q5_0 *) w, (const orc_block_q8_0 *) a);
        case ORC_Q5_1: return orc_vec_dot_q5_1_q8_1_avx2(n, (const orc_block_q5_1 *) w, (const orc_block_q8_1 *) a);
        case ORC_IQ4_NL: return orc_vec_dot_iq4_nl_q8_0_avx2(n, (const orc_block_iq4_nl *) w, (const orc_block_q8_0 *) a, 0);
        case ORC_MXFP4: return orc_vec_dot_mxfp4_q8_0_avx2(n, (const orc_block_mxfp4 *) w, (const orc_block_q8_0 *) a);
        case ORC_IQ4_XS: return orc_vec_dot_iq4_xs_q8_K_avx2(n, (const orc_block_iq4_xs *) w, (const orc_block_q8_K *) a);
        case ORC_TQ1_0: return orc_vec_dot_tq1_0_q8_K_avx2(n, (const orc_block_tq1_0 *) w, (const orc_block_q8_K *) a);
        case ORC_TQ2_0: return orc_vec_dot_tq2_0_q8_K_avx2(n, (const orc_block_tq2_0 *) w, (const orc_block_q8_K *) a);
        case ORC_IQ2_XXS: case ORC_IQ2_XS: case ORC_IQ2_S: case ORC_IQ3_XXS: case ORC_IQ3_S: return orc_vec_dot_iq_grid_q8_K_avx2(wtype, n, w, (const orc_block_q8_K *) a);
        case ORC_IQ1_S: return orc_vec_dot_iq1_s_q8_K_avx2(n, (const orc_block_iq1_s *) w, (const orc_block_q8_K *) a);
        case ORC_IQ1_M: return orc_vec_dot_iq1_m_q8_K_avx2(n, (const orc_block_iq1_m *) w, (const orc_block_q8_K *) a);
        case ORC_Q2_K: return orc_vec_dot_q2_K_q8_K_avx2(n, (const orc_block_q2_K *) w, (const orc_block_q8_K *) a);
        case ORC_Q3_K: return orc_vec_dot_q3_K_q8_K_avx2(n, (const orc_block_q3_K *) w, (const orc_block_q8_K *) a);
        case ORC_F16: {
            if (g_order == ORC_ORDER_AVX2) return orc_vec_dot_f16_avx2(n, (const uint16_t *) w, (const uint16_t *) a);          /* scalar branch of ggml_vec_dot_f16 (vec.cpp:264-): double accumulation */
            const uint16_t * x = (const uint16_t *) w, * y = (const uint16_t *) a;
            double s = 0.0;
            for (int64_t i = 0; i < n; i++) s += (double)(orc_fp16_to_fp32(x[i]) * orc_fp16_to_fp32(y[i]));
            return (float) s;
        }
        case ORC_F32: {          /* scalar branch of ggml_vec_dot_f32 (vec.cpp:10-): double accumulation */
            if (g_order == ORC_ORDER_AVX2) return orc_vec_dot_f32_avx2(n, (const float *) w, (const float *) a);
            const float * x = (const float *) w, * y = (const float *) a;
            double s = 0.0;
            for (int64_t i = 0; i < n; i++) s += (double)(x[i] * y[i]);
            return (float) s;
        }
    }
    return 0;
}

int orc_mul_mat(const orc_tensor * src0, const orc_tensor * src1, orc_tensor * dst) {
    const int64_t K = src0->ne[0];
    if (src1->ne[0] != K || src1->type != ORC_F32 || dst->type != ORC_F32) return -1;
    if (dst->ne[0] != src0->ne[1] || dst->ne[1] != src1->ne[1] || dst->ne[2] != src1->ne[2] || dst->ne[3] != src1->ne[3]) return -2;
    if (src1->ne[2] % src0->ne[2] || src1->ne[3] % src0->ne[3]) return -3;
    if (src0->nb[0] != orc_type_size(src0->type) || src1->nb[0] != 4) return -4;   /* rows themselves are dense */
    const int vt = vec_dot_type_of(src0->type);
    if (vt < 0 || K % orc_blck_size(vt)) return -5;

    const size_t rs = orc_row_size(vt, K);
    void * arow = malloc(rs);
    const int64_t r2 = src1->ne[2] / src0->ne[2], r3 = src1->ne[3] / src0->ne[3];
    for (int64_t i13 = 0; i13 < src1->ne[3]; i13++)
    for (int64_t i12 = 0; i12 < src1->ne[2]; i12++)
    for (int64_t i11 = 0; i11 < src1->ne[1]; i11++) {
        convert_row(vt, (const float *) tptr(src1, 0, i11, i12, i13), arow, K);
        /* llamafile_sgemm takes F16 / F32 src0 for n >= 2, k % 8 == 0, m % 4 == 0 (sgemm.cpp:3691, 488, 503-517) */
        const int tiny = g_order == ORC_ORDER_AVX2 && (src0->type == ORC_F16 || src0->type == ORC_F32) && src1->ne[1] >= 2 && K % 8 == 0 && src0->ne[1] % 4 == 0;
        for (int64_t i01 = 0; i01 < src0->ne[1]; i01++) {
            const void * w = tptr(src0, 0, i01, i12 / r2, i13 / r3);
            float v;
            if (tiny) v = src0->type == ORC_F16 ? tiny8_f16(K, (const uint16_t *) w, (const uint16_t *) arow) : tiny8_f32(K, (const float *) w, (const float *) arow);
            else if (src0->type == ORC_IQ4_NL && src1->ne[1] >= 2) v = orc_vec_dot_iq4_nl_q8_0_avx2(K, (const orc_block_iq4_nl *) w, (const orc_block_q8_0 *) arow, 1);   /* tinyBLAS_Q0_AVX (sgemm.cpp:3691, 4000-4013) */
            else v = vec_dot(src0->type, K, w, arow);
            *(float *) tptr(dst, i01, i11, i12, i13) = v;
        }
    }
    free(arow);
    return 0;
}

/* ggml-cpu.c:1432-1678: dst[:, id, tok] = as[:, :, ids[id, tok]]^T . b[:, id % ne11, tok] */
int orc_mul_mat_id(const orc_tensor * as, const orc_tensor * b, const orc_tensor * ids, orc_tensor * dst) {
    const int64_t K = as->ne[0], N = as->ne[1], n_as = as->ne[2];
    const int64_t n_used = ids->ne[0], n_tok = ids->ne[1];
    if (b->ne[0] != K || b->type != ORC_F32 || ids->type != ORC_I32 || dst->type != ORC_F32) return -1;
    if (dst->ne[0] != N || dst->ne[1] != n_used || dst->ne[2] != n_tok || b->ne[2] != n_tok) return -2;
    const int vt = vec_dot_type_of(as->type);
    if (vt < 0) return -5;
    void * arow = malloc(orc_row_size(vt, K));
    for (int64_t t = 0; t < n_tok; t++)
    for (int64_t id = 0; id < n_used; id++) {
        const int32_t e = *(const int32_t *) tptr(ids, id, t, 0, 0);
        if (e < 0 || e >= n_as) { free(arow); return -6; }
        convert_row(vt, (const float *) tptr(b, 0, id % b->ne[1], t, 0), arow, K);
        for (int64_t r = 0; r < N; r++)
            *(float *) tptr(dst, r, id, t, 0) = vec_dot(as->type, K, tptr(as, 0, r, e, 0), arow);
    }
    free(arow);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* normalisation, rope, softmax, elementwise                                                   */
/* ------------------------------------------------------------------------------------------ */
int orc_rms_norm(const orc_tensor * src, orc_tensor * dst, float eps) {
    if (src->type != ORC_F32 || dst->type != ORC_F32 || src->nb[0] != 4 || dst->nb[0] != 4) return -1;
    const int64_t n = src->ne[0];
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++)
    for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {
        const float * x = (const float *) tptr(src, 0, i1, i2, i3);
        float * y = (float *) tptr(dst, 0, i1, i2, i3);
        double sum = 0.0;
        for (int64_t i = 0; i < n; i++) sum += (double)(x[i] * x[i]);
        const float mean  = (float)(sum / (double) n);
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int64_t i = 0; i < n; i++) y[i] = x[i] * scale;
    }
    return 0;
}

/* ggml.c ggml_rope_yarn_corr_dim(s) */
static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}
static void rope_corr_dims(int n_dims, int n_ctx_orig, float freq_base, float beta_fast, float beta_slow, float dims[2]) {
    const float start = floorf(rope_corr_dim(n_dims, n_ctx_orig, beta_fast, freq_base));
    const float end   = ceilf (rope_corr_dim(n_dims, n_ctx_orig, beta_slow, freq_base));
    dims[0] = ORC_MAX(0, start);
    dims[1] = ORC_MIN(n_dims - 1, end);
}
static float rope_ramp(float low, float high, int i0) {
    const float y = (i0 / 2 - low) / ORC_MAX(0.001f, high - low);
    return 1 - ORC_MIN(1, ORC_MAX(0, y));
}
/* ops.cpp:5596-5611 */
static void rope_yarn(float theta_extrap, float freq_scale, const float corr[2], int64_t i0, float ext_factor, float mscale,
                      float * c, float * s) {
    const float theta_interp = freq_scale * theta_extrap;
    float theta = theta_interp;
    if (ext_factor != 0.0f) {
        const float mix = rope_ramp(corr[0], corr[1], (int) i0) * ext_factor;
        /* as the reference build compiles the two statements (gcc contracts a * b + c into one fma; found by enumerating the contraction forms against
         * libggml-cpu.so, bit for bit: tests/test_oracle_vs_reference.py::test_rope_yarn) */
        theta = fmaf(theta_interp, 1 - mix, theta_extrap * mix);
        mscale *= fmaf(0.1f, logf(1.0f / freq_scale), 1.0f);
    }
    *c = cosf(theta) * mscale;
    *s = sinf(theta) * mscale;
}

int orc_rope(const orc_tensor * src, const int32_t * pos, const float * ff, orc_tensor * dst, const orc_rope_params * p) {
    if (src->type != ORC_F32 || dst->type != ORC_F32 || src->nb[0] != 4 || dst->nb[0] != 4) return -1;
    if (p->mode != 0 && p->mode != 2) return -2;
    const int64_t ne0 = src->ne[0];
    const int n_dims = p->n_dims;
    if (n_dims > ne0 || (n_dims & 1)) return -3;
    const float theta_scale = powf(p->freq_base, -2.0f / n_dims);
    float corr[2];
    rope_corr_dims(n_dims, p->n_ctx_orig, p->freq_base, p->beta_fast, p->beta_slow, corr);
    float * cache = (float *) malloc((size_t) ne0 * sizeof(float));

    for (int64_t i3 = 0; i3 < src->ne[3]; i3++)
    for (int64_t i2 = 0; i2 < src->ne[2]; i2++) {                    /* sequence position */
        float theta = (float) pos[i2];                               /* iterated product, ops.cpp:5613-5628 */
        for (int64_t i0 = 0; i0 < ne0; i0 += 2) {
            const float f = ff ? ff[i0/2] : 1.0f;
            rope_yarn(theta / f, p->freq_scale, corr, i0, p->ext_factor, p->attn_factor, &cache[i0], &cache[i0 + 1]);
            theta *= theta_scale;
        }
        for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {                /* heads */
            const float * x = (const float *) tptr(src, 0, i1, i2, i3);
            float * y = (float *) tptr(dst, 0, i1, i2, i3);
            const int64_t off = p->mode == 0 ? 1 : n_dims / 2;
            for (int64_t i0 = 0; i0 < n_dims; i0 += 2) {
                const int64_t ic = p->mode == 0 ? i0 : i0 / 2;
                const float c = cache[i0], s = cache[i0 + 1];
                const float x0 = x[ic], x1 = x[ic + off];
                /* rotate_pairs (ops.cpp:5700-5718) as the reference build compiles it: gcc contracts each expression into one fma
                 * around the rounded x1 product (verified bit for bit against libggml-cpu.so) */
                y[ic]       = fmaf(x0, c, -(x1*s));
                y[ic + off] = fmaf(x0, s, x1*c);
            }
            for (int64_t i0 = n_dims; i0 < ne0; i0++) y[i0] = x[i0];
        }
    }
    free(cache);
    return 0;
}

/* one lane of the AVX2 ggml_v_expf (vec.h:1230-1267); every fma of the original is an fmaf here */
float orc_expf_avx2(float x) {
    const float r = 0x1.8p23f;
    const float z = fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = fmaf(-n, 0x1.7f7d1cp-20f, fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = f2u(z) << 23;
    const float k = u2f(e + f2u(1.0f));
    const int   c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = fmaf(fmaf(fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u,
                              fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)),
                         u, 0x1.ffffecp-1f * b);
    if (!c) return fmaf(j, k, k);
    const uint32_t g  = (n <= 0.0f) ? 0x82000000u : 0u;
    const float    s1 = u2f(g + 0x7f000000u);
    const float    s2 = u2f(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return fmaf(s2, j, s2) * s1;
}
float orc_silu_avx2(float x) { return x / (1.0f + orc_expf_avx2(0.0f - x)); }

int orc_soft_max(const orc_tensor * src, const orc_tensor * mask, orc_tensor * dst, float scale, float max_bias) {
    if (src->type != ORC_F32 || dst->type != ORC_F32 || max_bias != 0.0f) return -1;   /* ALiBi not on this path */
    if (mask && mask->type != ORC_F32 && mask->type != ORC_F16) return -2;
    const int64_t n = src->ne[0];
    float * wp = (float *) malloc((size_t) n * sizeof(float));
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++)
    for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {
        const float * sp = (const float *) tptr(src, 0, i1, i2, i3);
        float * dp = (float *) tptr(dst, 0, i1, i2, i3);
        for (int64_t i = 0; i < n; i++) wp[i] = sp[i] * scale;
        if (mask) {
            const char * mp = tptr(mask, 0, i1, i2 % mask->ne[2], i3 % mask->ne[3]);
            for (int64_t i = 0; i < n; i++)
                wp[i] += 1.0f * (mask->type == ORC_F16 ? orc_fp16_to_fp32(((const uint16_t *) mp)[i]) : ((const float *) mp)[i]);
        }
        float max = -INFINITY;
        for (int64_t i = 0; i < n; i++) max = ORC_MAX(max, wp[i]);
        /* ggml_vec_soft_max_f32 (vec.cpp:547-): groups of 8 through the polynomial, hsum in f32, total in double; tail expf */
        double sum = 0.0;
        int64_t i = 0;
        for (; i + 7 < n; i += 8) {
            float v[8];
            for (int l = 0; l < 8; l++) { v[l] = orc_expf_avx2(wp[i + l] - max); dp[i + l] = v[l]; }
            /* extractf128 add, movehl add, movehdup add */
            const float a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
            const float b0 = a0 + a2, b1 = a1 + a3;
            sum += (double)(b0 + b1);
        }
        for (; i < n; i++) { const float v = expf(wp[i] - max); dp[i] = v; sum += (double) v; }
        const float inv = (float)(1.0 / sum);
        for (int64_t l = 0; l < n; l++) dp[l] *= inv;
    }
    free(wp);
    return 0;
}

int orc_flash_attn_ext(const orc_tensor * q, const orc_tensor * k, const orc_tensor * v, const orc_tensor * mask, orc_tensor * dst, float scale) {
    if (q->type != ORC_F32 || dst->type != ORC_F32 || k->type != v->type || (k->type != ORC_F16 && k->type != ORC_Q8_0)) return -1;
    if (mask && mask->type != ORC_F16) return -2;
    const int64_t DK = k->ne[0], DV = v->ne[0], N = q->ne[1], H = q->ne[2], B = q->ne[3], n_kv = k->ne[1];
    if (q->ne[0] != DK || DK % 32 || DV % 32 || H % k->ne[2] || H % v->ne[2] || B % k->ne[3] || dst->ne[0] != DV || dst->ne[1] != H || dst->ne[2] != N) return -3;
    const int64_t rk2 = H / k->ne[2], rk3 = B / k->ne[3], rv2 = H / v->ne[2], rv3 = B / v->ne[3];
    uint16_t * Qh = (uint16_t *) malloc((size_t) DK * 2 + (size_t)(DK / 32) * sizeof(orc_block_q8_0));
    orc_block_q8_0 * Qq = (orc_block_q8_0 *)(Qh + DK);
    uint16_t * VKQ16 = (uint16_t *) malloc((size_t) DV * 2);
    float * VKQ32 = (float *) malloc((size_t) DV * 8), * V32 = VKQ32 + DV;
    for (int64_t iq3 = 0; iq3 < B; iq3++)
    for (int64_t iq2 = 0; iq2 < H; iq2++)
    for (int64_t iq1 = 0; iq1 < N; iq1++) {
        const float * pq = (const float *) tptr(q, 0, iq1, iq2, iq3);
        if (k->type == ORC_F16) for (int64_t i = 0; i < DK; i++) Qh[i] = orc_fp32_to_fp16(pq[i]);      /* from_float of the vec_dot_type */
        else orc_quantize_row_q8_0(pq, Qq, DK);
        float S = 0.0f, M = -INFINITY;
        if (v->type == ORC_F16) memset(VKQ16, 0, (size_t) DV * 2); else memset(VKQ32, 0, (size_t) DV * 4);
        const uint16_t * mp = mask ? (const uint16_t *) tptr(mask, 0, iq1, iq2 % mask->ne[2], iq3 % mask->ne[3]) : NULL;
        for (int64_t ic = 0; ic < n_kv; ic++) {
            const float mv = mp ? 1.0f * orc_fp16_to_fp32(mp[ic]) : 0.0f;
            if (mv == -INFINITY) continue;
            const char * kd = tptr(k, 0, ic, iq2 / rk2, iq3 / rk3);
            float s = k->type == ORC_F16 ? orc_vec_dot_f16_avx2(DK, (const uint16_t *) kd, Qh)
                                         : orc_vec_dot_q8_0_q8_0_avx2(DK, (const orc_block_q8_0 *) kd, Qq);
            s = s * scale;
            s += mv;
            const float Mold = M;
            float ms = 1.0f, vs = 1.0f;
            const char * vd = tptr(v, 0, ic, iq2 / rv2, iq3 / rv3);
            if (v->type == ORC_F16) {
                if (s > M) {
                    M = s; ms = expf(Mold - M);
                    for (int64_t d = 0; d < DV; d++) VKQ16[d] = orc_fp32_to_fp16(orc_fp16_to_fp32(VKQ16[d]) * ms);            /* ggml_vec_scale_f16 */
                } else vs = expf(s - M);
                for (int64_t d = 0; d < DV; d++)                                                                                 /* ggml_vec_mad_f16: F32Cx8 fmadd */
                    VKQ16[d] = orc_fp32_to_fp16(fmaf(orc_fp16_to_fp32(((const uint16_t *) vd)[d]), vs, orc_fp16_to_fp32(VKQ16[d])));
            } else {
                if (s > M) {
                    M = s; ms = expf(Mold - M);
                    for (int64_t d = 0; d < DV; d++) VKQ32[d] *= ms;
                } else vs = expf(s - M);
                orc_dequantize_row_q8_0((const orc_block_q8_0 *) vd, V32, DV);
                for (int64_t d = 0; d < DV; d++) VKQ32[d] = fmaf(V32[d], vs, VKQ32[d]);                                          /* ggml_vec_mad_f32 */
            }
            S = S * ms + vs;
        }
        if (v->type == ORC_F16) for (int64_t d = 0; d < DV; d++) VKQ32[d] = orc_fp16_to_fp32(VKQ16[d]);
        const float S_inv = S == 0.0f ? 0.0f : 1.0f / S;
        float * dp = (float *) tptr(dst, 0, iq2, iq1, iq3);
        for (int64_t d = 0; d < DV; d++) dp[d] = VKQ32[d] * S_inv;
    }
    free(Qh); free(VKQ16); free(VKQ32);
    return 0;
}

int orc_diag_mask_inf(const orc_tensor * src, orc_tensor * dst, int n_past) {
    if (src->type != ORC_F32 || dst->type != ORC_F32) return -1;
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++)
    for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t j = 0; j < src->ne[1]; j++)
    for (int64_t i = 0; i < src->ne[0]; i++) {
        const float v = *(const float *) tptr(src, i, j, i2, i3);
        *(float *) tptr(dst, i, j, i2, i3) = (i > n_past + j) ? -INFINITY : v;
    }
    return 0;
}

int orc_scale(const orc_tensor * src, orc_tensor * dst, float s, float b) {
    if (src->type != ORC_F32 || dst->type != ORC_F32) return -1;
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t i1 = 0; i1 < src->ne[1]; i1++) for (int64_t i0 = 0; i0 < src->ne[0]; i0++) {
        const float v = *(const float *) tptr(src, i0, i1, i2, i3);
        /* ggml_vec_scale_f32 when b == 0, ggml_vec_mad1_f32 otherwise */
        *(float *) tptr(dst, i0, i1, i2, i3) = (b == 0.0f) ? v * s : v * s + b;
    }
    return 0;
}

int orc_silu(const orc_tensor * src, orc_tensor * dst) {
    if (src->type != ORC_F32 || dst->type != ORC_F32 || src->nb[0] != 4 || dst->nb[0] != 4) return -1;
    const int64_t n = src->ne[0];
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {
        const float * x = (const float *) tptr(src, 0, i1, i2, i3);
        float * y = (float *) tptr(dst, 0, i1, i2, i3);
        const int64_t nv = n & ~(int64_t) 7;                        /* vector body: indices < (n & ~7) */
        for (int64_t i = 0;  i < nv; i++) y[i] = orc_silu_avx2(x[i]);
        for (int64_t i = nv; i < n;  i++) y[i] = x[i] / (1.0f + expf(-x[i]));
    }
    return 0;
}

static int binary_op(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst, int op) {
    if (a->type != ORC_F32 || b->type != ORC_F32 || dst->type != ORC_F32) return -1;
    for (int d = 0; d < 4; d++) if (a->ne[d] % b->ne[d] || dst->ne[d] != a->ne[d]) return -2;
    for (int64_t i3 = 0; i3 < a->ne[3]; i3++) for (int64_t i2 = 0; i2 < a->ne[2]; i2++)
    for (int64_t i1 = 0; i1 < a->ne[1]; i1++) for (int64_t i0 = 0; i0 < a->ne[0]; i0++) {
        const float x = *(const float *) tptr(a, i0, i1, i2, i3);
        const float y = *(const float *) tptr(b, i0 % b->ne[0], i1 % b->ne[1], i2 % b->ne[2], i3 % b->ne[3]);
        *(float *) tptr(dst, i0, i1, i2, i3) = op == 0 ? x + y : op == 1 ? x * y : x / y;
    }
    return 0;
}
int orc_add(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst) { return binary_op(a, b, dst, 0); }
int orc_mul(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst) { return binary_op(a, b, dst, 1); }
int orc_div(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst) { return binary_op(a, b, dst, 2); }

/* ggml_compute_forward_sum_rows_f32 (ops.cpp:1451-1482) with ggml_vec_sum_f32 (vec.h:1510-1520): sequential, double accumulator */
int orc_sum_rows(const orc_tensor * src, orc_tensor * dst) {
    if (src->type != ORC_F32 || dst->type != ORC_F32 || dst->ne[0] != 1) return -1;
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++) for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {
        double sum = 0.0;
        for (int64_t i0 = 0; i0 < src->ne[0]; i0++) sum += (double) *(const float *) tptr(src, i0, i1, i2, i3);
        *(float *) tptr(dst, 0, i1, i2, i3) = (float) sum;
    }
    return 0;
}
/* ggml_compute_forward_top_k_f32 (ops.cpp:8057-8094): k largest in descending order (ties: lower index first), first two swapped */
int orc_top_k(const orc_tensor * src, orc_tensor * dst) {
    if (src->type != ORC_F32 || dst->type != ORC_I32 || dst->ne[0] > src->ne[0]) return -1;
    const int64_t n = src->ne[0], k = dst->ne[0];
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++) for (int64_t i1 = 0; i1 < src->ne[1]; i1++) {
        int32_t * out = (int32_t *) tptr(dst, 0, i1, i2, i3);
        for (int64_t j = 0; j < k; j++) {
            int64_t bi = -1; float best = 0;
            for (int64_t i = 0; i < n; i++) {
                int taken = 0;
                for (int64_t t = 0; t < j; t++) taken |= out[t] == (int32_t) i;
                const float v = *(const float *) tptr(src, i, i1, i2, i3);
                if (!taken && (bi < 0 || v > best)) { best = v; bi = i; }
            }
            out[j] = (int32_t) bi;
        }
        if (k > 1) { const int32_t t = out[0]; out[0] = out[1]; out[1] = t; }
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* KV-cache writes and gathers                                                                 */
/* ------------------------------------------------------------------------------------------ */
int orc_set_rows(const orc_tensor * src, const orc_tensor * idx, orc_tensor * dst) {
    if (src->type != ORC_F32 || (dst->type != ORC_F16 && dst->type != ORC_F32)) return -1;
    if (idx->type != ORC_I32 && idx->type != ORC_I64) return -2;
    if (dst->ne[0] != src->ne[0] || dst->ne[2] != src->ne[2] || dst->ne[3] != src->ne[3]) return -3;
    if (idx->ne[0] != src->ne[1] || src->ne[2] % idx->ne[1] || src->ne[3] % idx->ne[2]) return -4;
    for (int64_t i3 = 0; i3 < src->ne[3]; i3++) for (int64_t i2 = 0; i2 < src->ne[2]; i2++)
    for (int64_t i = 0; i < src->ne[1]; i++) {
        const char * ip = tptr(idx, i, i2 % idx->ne[1], i3 % idx->ne[2], 0);
        const int64_t r = idx->type == ORC_I32 ? *(const int32_t *) ip : *(const int64_t *) ip;
        if (r < 0 || r >= dst->ne[1]) return -5;
        const float * x = (const float *) tptr(src, 0, i, i2, i3);
        char * y = tptr(dst, 0, r, i2, i3);
        for (int64_t c = 0; c < src->ne[0]; c++) {
            if (dst->type == ORC_F16) ((uint16_t *) y)[c] = orc_fp32_to_fp16(x[c]);
            else                      ((float    *) y)[c] = x[c];
        }
    }
    return 0;
}

/* element e (row-major over ne) of src goes to element e of dst: ggml_compute_forward_dup semantics */
int orc_cpy(const orc_tensor * src, orc_tensor * dst) {
    const int64_t n = src->ne[0]*src->ne[1]*src->ne[2]*src->ne[3];
    if (n != dst->ne[0]*dst->ne[1]*dst->ne[2]*dst->ne[3]) return -1;
    if ((src->type != ORC_F32 && src->type != ORC_F16) || (dst->type != ORC_F32 && dst->type != ORC_F16)) return -2;
    for (int64_t e = 0; e < n; e++) {
        int64_t r = e;
        const int64_t s0 = r % src->ne[0]; r /= src->ne[0];
        const int64_t s1 = r % src->ne[1]; r /= src->ne[1];
        const int64_t s2 = r % src->ne[2]; const int64_t s3 = r / src->ne[2];
        r = e;
        const int64_t d0 = r % dst->ne[0]; r /= dst->ne[0];
        const int64_t d1 = r % dst->ne[1]; r /= dst->ne[1];
        const int64_t d2 = r % dst->ne[2]; const int64_t d3 = r / dst->ne[2];
        const char * sp = tptr(src, s0, s1, s2, s3);
        char * dp = tptr(dst, d0, d1, d2, d3);
        if (src->type == dst->type) { memcpy(dp, sp, orc_type_size(src->type)); continue; }
        if (src->type == ORC_F32) *(uint16_t *) dp = orc_fp32_to_fp16(*(const float *) sp);
        else                      *(float *) dp    = orc_fp16_to_fp32(*(const uint16_t *) sp);
    }
    return 0;
}

int orc_get_rows(const orc_tensor * src, const orc_tensor * idx, orc_tensor * dst) {
    if (idx->type != ORC_I32 || dst->type != ORC_F32 || dst->ne[0] != src->ne[0]) return -1;
    for (int64_t i12 = 0; i12 < idx->ne[2]; i12++) for (int64_t i11 = 0; i11 < idx->ne[1]; i11++)
    for (int64_t i10 = 0; i10 < idx->ne[0]; i10++) {
        const int32_t r = *(const int32_t *) tptr(idx, i10, i11, i12, 0);
        if (r < 0 || r >= src->ne[1]) return -2;
        orc_dequantize_row(src->type, tptr(src, 0, r, i11, i12), (float *) tptr(dst, 0, i10, i11, i12), src->ne[0]);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* whole-model forward (SURVEY.md section 3.3 node order)                                      */
/* ------------------------------------------------------------------------------------------ */
static orc_tensor T2(int type, void * data, int64_t n0, int64_t n1) {
    orc_tensor t; t.type = type; t.data = data;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = 1; t.ne[3] = 1;
    t.nb[0] = orc_type_size(type); t.nb[1] = orc_row_size(type, n0); t.nb[2] = t.nb[1]*(size_t)n1; t.nb[3] = t.nb[2];
    return t;
}
static void linear(const orc_weight * w, int64_t K, int64_t N, float * x, int64_t qlen, float * y, const float * bias) {
    orc_tensor W = T2(w->type, (void *) w->data, K, N), X = T2(ORC_F32, x, K, qlen), Y = T2(ORC_F32, y, N, qlen);
    orc_mul_mat(&W, &X, &Y);
    if (bias) for (int64_t t = 0; t < qlen; t++) for (int64_t i = 0; i < N; i++) y[t*N + i] += bias[i];
}
static void norm_mul(const float * w, float * x, float * y, int64_t H, int64_t qlen, float eps) {
    orc_tensor X = T2(ORC_F32, x, H, qlen), Y = T2(ORC_F32, y, H, qlen);
    orc_rms_norm(&X, &Y, eps);                                  /* RMS_NORM node */
    for (int64_t t = 0; t < qlen; t++) for (int64_t i = 0; i < H; i++) y[t*H + i] *= w[i];   /* MUL node */
}

int orc_llama_forward(orc_llama_model * m, const int32_t * tokens, int qlen, int n_past, float * logits) {
    const orc_llama_config * c = &m->cfg;
    const int64_t H = c->hidden, hd = c->head_dim, nh = c->n_head, nkv = c->n_kv_head, F = c->ffn, V = c->vocab;
    const int64_t QD = nh*hd, KD = nkv*hd, n_kv = n_past + qlen, ML = c->max_len;
    if (n_kv > ML) return -1;

    float * x   = (float *) malloc(sizeof(float) * (size_t)(H*qlen));
    float * xn  = (float *) malloc(sizeof(float) * (size_t)(H*qlen));
    float * q   = (float *) malloc(sizeof(float) * (size_t)(QD*qlen));
    float * k   = (float *) malloc(sizeof(float) * (size_t)(KD*qlen));
    float * v   = (float *) malloc(sizeof(float) * (size_t)(KD*qlen));
    float * att = (float *) malloc(sizeof(float) * (size_t)(QD*qlen));
    float * o   = (float *) malloc(sizeof(float) * (size_t)(H*qlen));
    float * g   = (float *) malloc(sizeof(float) * (size_t)(F*qlen));
    float * u   = (float *) malloc(sizeof(float) * (size_t)(F*qlen));
    float * sc  = (float *) malloc(sizeof(float) * (size_t)(n_kv*qlen*nh));
    float * ctx = (float *) malloc(sizeof(float) * (size_t)(hd*qlen*nh));
    int32_t * pos = (int32_t *) malloc(sizeof(int32_t) * (size_t) qlen);
    for (int t = 0; t < qlen; t++) pos[t] = n_past + t;

    { /* Embedding::forward -> GET_ROWS */
        orc_tensor E = T2(m->tok_embd.type, (void *) m->tok_embd.data, H, V), I = T2(ORC_I32, (void *) tokens, qlen, 1), X = T2(ORC_F32, x, H, qlen);
        if (orc_get_rows(&E, &I, &X)) return -2;
    }
    orc_rope_params rp = { (int32_t) hd, c->rope_mode, 0, c->rope_theta, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f };

    for (int il = 0; il < c->n_layer; il++) {
        const orc_llama_layer * L = &m->layers[il];
        uint16_t * kc = m->k_cache + (size_t) il * (size_t)(ML*KD);
        uint16_t * vc = m->v_cache + (size_t) il * (size_t)(ML*KD);

        norm_mul(L->attn_norm, x, xn, H, qlen, c->rms_eps);
        linear(&L->wq, H, QD, xn, qlen, q, c->qkv_bias ? L->bq : NULL);
        linear(&L->wk, H, KD, xn, qlen, k, c->qkv_bias ? L->bk : NULL);
        linear(&L->wv, H, KD, xn, qlen, v, c->qkv_bias ? L->bv : NULL);

        { /* rope in place on k then q: [hd, heads, qlen] */
            orc_tensor Kt = { ORC_F32, {hd, nkv, qlen, 1}, {4, 4*(size_t)hd, 4*(size_t)KD, 4*(size_t)(KD*qlen)}, k };
            orc_tensor Qt = { ORC_F32, {hd, nh,  qlen, 1}, {4, 4*(size_t)hd, 4*(size_t)QD, 4*(size_t)(QD*qlen)}, q };
            orc_rope(&Kt, pos, NULL, &Kt, &rp);
            orc_rope(&Qt, pos, NULL, &Qt, &rp);
        }
        /* save_to_cache: V transposed CPY F32->F16 into [KD][ML] at column n_past; K SET_ROWS into [ML][KD] */
        for (int t = 0; t < qlen; t++) for (int64_t i = 0; i < KD; i++) {
            vc[i*ML + (n_past + t)]  = orc_fp32_to_fp16(v[t*KD + i]);
            kc[(n_past + t)*KD + i]  = orc_fp32_to_fp16(k[t*KD + i]);
        }
        { /* scores = K^T Q (F16 x F16(q)), scale, causal mask, softmax, ctx = V P (F16 x F16(p)) */
            orc_tensor Kv = { ORC_F16, {hd, n_kv, nkv, 1}, {2, 2*(size_t)KD, 2*(size_t)hd, 2*(size_t)(KD*ML)}, kc };
            orc_tensor Qv = { ORC_F32, {hd, qlen, nh, 1},  {4, 4*(size_t)QD, 4*(size_t)hd, 4*(size_t)(QD*qlen)}, q };
            orc_tensor S  = { ORC_F32, {n_kv, qlen, nh, 1}, {4, 4*(size_t)n_kv, 4*(size_t)(n_kv*qlen), 4*(size_t)(n_kv*qlen*nh)}, sc };
            orc_mul_mat(&Kv, &Qv, &S);
            orc_scale(&S, &S, 1.0f / sqrtf((float) hd), 0.0f);
            orc_diag_mask_inf(&S, &S, n_past);
            orc_soft_max(&S, NULL, &S, 1.0f, 0.0f);
            orc_tensor Vv = { ORC_F16, {n_kv, hd, nkv, 1}, {2, 2*(size_t)ML, 2*(size_t)(ML*hd), 2*(size_t)(ML*KD)}, vc };
            orc_tensor C  = { ORC_F32, {hd, qlen, nh, 1}, {4, 4*(size_t)hd, 4*(size_t)(hd*qlen), 4*(size_t)(hd*qlen*nh)}, ctx };
            orc_mul_mat(&Vv, &S, &C);
            /* permute(0,2,1,3) + cont -> [hd, nh, qlen] */
            for (int t = 0; t < qlen; t++) for (int64_t h = 0; h < nh; h++)
                memcpy(att + t*QD + h*hd, ctx + (h*qlen + t)*hd, sizeof(float) * (size_t) hd);
        }
        linear(&L->wo, QD, H, att, qlen, o, NULL);
        for (int64_t i = 0; i < H*qlen; i++) x[i] = o[i] + x[i];                 /* ADD residual */

        norm_mul(L->ffn_norm, x, xn, H, qlen, c->rms_eps);
        linear(&L->wgate, H, F, xn, qlen, g, NULL);
        { orc_tensor G = T2(ORC_F32, g, F, qlen); orc_silu(&G, &G); }
        linear(&L->wup, H, F, xn, qlen, u, NULL);
        for (int64_t i = 0; i < F*qlen; i++) g[i] = g[i] * u[i];
        linear(&L->wdown, F, H, g, qlen, o, NULL);
        for (int64_t i = 0; i < H*qlen; i++) x[i] = o[i] + x[i];
    }
    /* LMFinalSteps: last token -> norm -> lm_head */
    norm_mul(m->out_norm, x + (size_t)(qlen - 1)*H, xn, H, 1, c->rms_eps);
    linear(&m->lm_head, H, V, xn, 1, logits, NULL);

    free(x); free(xn); free(q); free(k); free(v); free(att); free(o); free(g); free(u); free(sc); free(ctx); free(pos);
    return 0;
}
