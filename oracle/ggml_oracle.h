/*
 * oracle/ggml_oracle.h -- TEST INFRASTRUCTURE.  NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the reference's CPU algorithm for the transformer forward hot path
 * (SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this; the HIP path never calls into it.
 *
 * Parity pin: the reference ships no tests / golden vectors, so this
 * restatement is pinned against the reference ITSELF: oracle/_ref/libggml-cpu.so is compiled
 * from the sources under /root/reference by oracle/Makefile and driven through
 * oracle/ref_ops.c; tests/test_oracle_vs_reference.py checks every function here against it,
 * and tests/golden/ holds vectors generated from it (tests/golden/make_golden.py).
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#ifndef GGML_ORACLE_H
#define GGML_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* numeric values equal the reference's enum ggml_type (ggml/include/ggml.h:386-428) */
enum orc_type {
    ORC_F32 = 0, ORC_F16 = 1, ORC_Q4_0 = 2, ORC_Q4_1 = 3, ORC_Q5_0 = 6, ORC_Q5_1 = 7, ORC_Q8_0 = 8, ORC_Q8_1 = 9, ORC_Q2_K = 10, ORC_Q3_K = 11, ORC_Q4_K = 12, ORC_Q5_K = 13, ORC_Q6_K = 14, ORC_Q8_K = 15, ORC_IQ4_NL = 20, ORC_IQ2_XXS = 16, ORC_IQ2_XS = 17, ORC_IQ3_XXS = 18, ORC_IQ1_S = 19, ORC_IQ1_M = 29, ORC_IQ3_S = 21, ORC_IQ2_S = 22, ORC_IQ4_XS = 23, ORC_TQ1_0 = 34, ORC_TQ2_0 = 35, ORC_I32 = 26, ORC_I64 = 27, ORC_MXFP4 = 39,
};

/* a strided 4-D tensor view: same meaning as ggml_tensor {type, ne, nb, data} (ggml.h:656-688) */
typedef struct orc_tensor {
    int32_t type;
    int64_t ne[4];
    size_t  nb[4];
    void *  data;
} orc_tensor;

/* ---- formats (ggml/src/ggml-common.h:170-175, 177-189, 219-224, 226-237, 295-306, 338-343) ---- */
#define ORC_QK    32
#define ORC_QK_K  256
#pragma pack(push, 1)
typedef struct { uint16_t d; uint8_t qs[16]; }                         orc_block_q4_0;   /* 18 B  */
typedef struct { uint16_t d; int8_t  qs[32]; }                         orc_block_q8_0;   /* 34 B  */
typedef struct { uint16_t d; uint16_t m; uint8_t qs[16]; }             orc_block_q4_1;   /* 20 B: w = nib * d + m */
typedef struct { uint16_t d; uint16_t s; int8_t qs[32]; }              orc_block_q8_1;   /* 36 B: s = d * sum(qs) */
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qs[128]; } orc_block_q4_K; /* 144 B */
typedef struct { uint16_t d; uint16_t dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; } orc_block_q5_K; /* 176 B: Q4_K + a fifth bit (ggml-common.h:308-321) */
typedef struct { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; }                  orc_block_q6_K; /* 210 B: 6-bit quants - 32, int8 scale per 16 (ggml-common.h:323-336) */
typedef struct { float d; int8_t qs[256]; int16_t bsums[16]; }         orc_block_q8_K;   /* 292 B */
/* the other 32-weight and k-quant formats stock model files carry (ggml-common.h:190-216, 262-288, 415-419) */
typedef struct { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; }              orc_block_q5_0;   /* 22 B: w = ((nib | bit << 4) - 16) * d */
typedef struct { uint16_t d; uint16_t m; uint8_t qh[4]; uint8_t qs[16]; }  orc_block_q5_1;   /* 24 B: w = (nib | bit << 4) * d + m */
typedef struct { uint16_t d; uint8_t qs[16]; }                             orc_block_iq4_nl; /* 18 B: w = kvalues_iq4nl[nib] * d */
typedef struct { uint16_t d; uint16_t scales_h; uint8_t scales_l[4]; uint8_t qs[128]; } orc_block_iq4_xs; /* 136 B: 256 weights, w = kvalues_iq4nl[nib] * d * (ls - 32), ls = 6-bit scale per 32 (ggml-common.h:421-427) */
/* the grid formats (ggml-common.h:346-390): 256 weights = 8 sub-blocks of 32; a weight is +-(a codebook magnitude) * d * (a small odd scale) * const; codebooks: ../chatllm.cpp_amd/csrc/iq_grids.h */
typedef struct { uint16_t d; uint16_t qs[32]; } orc_block_iq2_xxs;                                              /* 66 B: per 32: 4 grid bytes | 4 x 7 sign bits + 4-bit scale */
typedef struct { uint16_t d; uint16_t qs[32]; uint8_t scales[8]; } orc_block_iq2_xs;                            /* 74 B: per 8: 9-bit grid index | 7 sign bits; a 4-bit scale per 16 */
typedef struct { uint16_t d; uint8_t qs[64]; uint8_t qh[8]; uint8_t scales[8]; } orc_block_iq2_s;              /* 82 B: qs[0..31] grid index low bytes, qs[32..63] sign bytes, qh 2 high bits per 8 */
typedef struct { uint16_t d; uint8_t qs[96]; } orc_block_iq3_xxs;                                               /* 98 B: qs[0..63] grid indices (4 weights each), then per 32: 4 x 7 sign bits + 4-bit scale */
typedef struct { uint16_t d; uint8_t qs[64]; uint8_t qh[8]; uint8_t signs[32]; uint8_t scales[4]; } orc_block_iq3_s;   /* 110 B: 9-bit grid indices (qs | qh bit), sign bytes, a 4-bit scale per 32 */
typedef struct { uint16_t d; uint8_t qs[32]; uint16_t qh[8]; } orc_block_iq1_s;                                  /* 50 B: 11-bit grid index per 8 (qs | 3 bits of qh), per 32: 3-bit scale, delta sign (ggml-common.h:392-398) */
typedef struct { uint8_t qs[32]; uint8_t qh[16]; uint8_t scales[8]; } orc_block_iq1_m;                          /* 56 B: per 8: index (qs | 3 bits of a qh nibble), delta sign (its 4th bit); 3-bit scale per 16; fp16 d in the scales' top nibbles */
typedef struct { uint8_t qs[48]; uint8_t qh[4]; uint16_t d; } orc_block_tq1_0;     /* 54 B: 256 ternary weights, 5 per byte of qs (base 3), 4 per byte of qh; w = (trit - 1) * d (ggml-common.h:241-249) */
typedef struct { uint8_t qs[64]; uint16_t d; } orc_block_tq2_0;                    /* 66 B: 256 ternary weights, 2 bits each; w = (q - 1) * d (ggml-common.h:251-256) */
typedef struct { uint8_t e; uint8_t qs[16]; }                              orc_block_mxfp4;  /* 17 B: w = kvalues_mxfp4[nib] * 2^(e - 128) */
typedef struct { uint8_t scales[16]; uint8_t qs[64]; uint16_t d; uint16_t dmin; }             orc_block_q2_K;   /* 84 B  */
typedef struct { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; }         orc_block_q3_K;   /* 110 B */
#pragma pack(pop)

size_t orc_type_size(int type);   /* bytes per block */
int    orc_blck_size(int type);   /* elements per block */
size_t orc_row_size(int type, int64_t ne);

/* fp16 <-> fp32, IEEE round-to-nearest-even (what F16C does; ggml-cpu/simd-mappings.h) */
float    orc_fp16_to_fp32(uint16_t h);
uint16_t orc_fp32_to_fp16(float f);

/* ---- activation quantizers ---- */
/* x86 AVX2 branch: id = 127/amax, round-half-even   (ggml-cpu/arch/x86/quants.c:290-386) */
void orc_quantize_row_q8_0(const float * x, orc_block_q8_0 * y, int64_t k);
/* portable branch: id = 1/d, roundf (half away)      (ggml/src/ggml-quants.c:199-222)      */
void orc_quantize_row_q8_0_ref(const float * x, orc_block_q8_0 * y, int64_t k);
/* x86 AVX2 branch of quantize_row_q8_1: as Q8_0 plus s = fp16(d * sum q) with the UNROUNDED d
 * (ggml-cpu/arch/x86/quants.c:388-480; portable: ggml-quants.c:225-258)                    */
void orc_quantize_row_q8_1(const float * x, orc_block_q8_1 * y, int64_t k);
/* (ggml/src/ggml-quants.c:2555-2592, nearest_int :444)                                      */
void orc_quantize_row_q8_K(const float * x, orc_block_q8_K * y, int64_t k);

/* ---- weight quantizers: the reference's from_float_ref, what the loader's re-quantization calls (src/chat.cpp:1246-1279, src/layers.cpp:358-373) ---- */
void orc_quantize_row_q4_0_ref(const float * x, orc_block_q4_0 * y, int64_t k);      /* ggml-quants.c:36-71 */
void orc_quantize_row_q4_1_ref(const float * x, orc_block_q4_1 * y, int64_t k);      /* ggml-quants.c:73-108 */
void orc_quantize_row_q4_K_ref(const float * x, orc_block_q4_K * y, int64_t k);      /* ggml-quants.c:1280-1350, make_qkx2_quants :622-702 */
int  orc_quantize_row_ref(int type, const float * x, void * y, int64_t k);           /* Q8_0, Q4_0, Q4_1, Q5_0, Q5_1, Q4_K, F16; -1: type not restated */

/* ---- weight dequantizers (ggml/src/ggml-quants.c:307-325, 401-414, 1352-1373, 703-711) ---- */
void orc_dequantize_row_q4_0(const orc_block_q4_0 * x, float * y, int64_t k);
void orc_dequantize_row_q8_0(const orc_block_q8_0 * x, float * y, int64_t k);
void orc_dequantize_row_q4_1(const orc_block_q4_1 * x, float * y, int64_t k);      /* ggml-quants.c:327-345 */
void orc_dequantize_row_q4_K(const orc_block_q4_K * x, float * y, int64_t k);
void orc_dequantize_row_q5_K(const orc_block_q5_K * x, float * y, int64_t k);      /* ggml-quants.c:1554-1579 */
void orc_dequantize_row_q6_K(const orc_block_q6_K * x, float * y, int64_t k);      /* ggml-quants.c:1762-1791 */
void orc_dequantize_row_q5_0(const orc_block_q5_0 * x, float * y, int64_t k);      /* ggml-quants.c:348-372 */
void orc_dequantize_row_q5_1(const orc_block_q5_1 * x, float * y, int64_t k);      /* ggml-quants.c:374-399 */
void orc_dequantize_row_mxfp4(const orc_block_mxfp4 * x, float * y, int64_t k);    /* ggml-quants.c:417-436 */
void orc_dequantize_row_q2_K(const orc_block_q2_K * x, float * y, int64_t k);      /* ggml-quants.c:784-815 */
void orc_dequantize_row_q3_K(const orc_block_q3_K * x, float * y, int64_t k);      /* ggml-quants.c:1128-1176 */
void orc_dequantize_row_iq4_nl(const orc_block_iq4_nl * x, float * y, int64_t k);  /* ggml-quants.c:2512-2528 */
void orc_dequantize_row_iq4_xs(const orc_block_iq4_xs * x, float * y, int64_t k);  /* ggml-quants.c:2530-2551 */
void orc_dequantize_row_iq_grid(int type, const void * x, float * y, int64_t k);   /* dequantize_row_iq2_xxs / _iq2_xs / _iq2_s / _iq3_xxs / _iq3_s (ggml-quants.c:2275-2460) */
void orc_dequantize_row_tq1_0(const orc_block_tq1_0 * x, float * y, int64_t k);    /* ggml-quants.c:2215-2252 */
void orc_dequantize_row_tq2_0(const orc_block_tq2_0 * x, float * y, int64_t k);    /* ggml-quants.c:2254-2271 */
void orc_dequantize_row(int type, const void * x, float * y, int64_t k);
void orc_quantize_row_q5_0_ref(const float * x, orc_block_q5_0 * y, int64_t k);      /* ggml-quants.c:110-152 */
void orc_quantize_row_q5_1_ref(const float * x, orc_block_q5_1 * y, int64_t k);      /* ggml-quants.c:154-197 */

/* ---- block dot products (ggml-cpu/quants.c:115-150, 305-333, 550-623) ----
 * isums (optional): per-block exact integer sums (tier T0): for Q4_0/Q8_0 one int32 per 32-block;
 * for Q4_K two int32 per super-block { sum_s sc_s*dot_s , sum_s m_s*bsum_s }. */
float orc_vec_dot_q4_0_q8_0(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y, int32_t * isums);
float orc_vec_dot_q8_0_q8_0(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y, int32_t * isums);
/* ggml-cpu/quants.c:152-186: sum_b (d_w d_a) * sum_j nib_j a_j + m_w * s_a */
float orc_vec_dot_q4_1_q8_1(int64_t n, const orc_block_q4_1 * x, const orc_block_q8_1 * y, int32_t * isums);
float orc_vec_dot_q4_K_q8_K(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y, int32_t * isums);

/* ---- the same dot products in the order of the x86 AVX2 branches (arch/x86/quants.c:543-577, 701-760, 1012-1040,
 * 1742-1822; vec.cpp:11-, 264-): bit-identical to the reference build (libggml-cpu.so, -march=x86-64-v3).
 * orc_set_order() selects which restatement mul_mat / mul_mat_id / the whole-model forward use (default: AVX2). */
enum { ORC_ORDER_GENERIC = 0, ORC_ORDER_AVX2 = 1 };
void  orc_set_order(int order);
int   orc_get_order(void);
float orc_vec_dot_q4_0_q8_0_avx2(int64_t n, const orc_block_q4_0 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q8_0_q8_0_avx2(int64_t n, const orc_block_q8_0 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q4_1_q8_1_avx2(int64_t n, const orc_block_q4_1 * x, const orc_block_q8_1 * y);
float orc_vec_dot_q4_K_q8_K_avx2(int64_t n, const orc_block_q4_K * x, const orc_block_q8_K * y);
/* Q5_K / Q6_K: only the x86 AVX2 order is restated (arch/x86/quants.c:1916-2030, 2130-2225): 8 lane accumulators, one fma per super-block;
 * Q5_K's mins go through one scalar chain `summs` */
float orc_vec_dot_q5_K_q8_K_avx2(int64_t n, const orc_block_q5_K * x, const orc_block_q8_K * y);
float orc_vec_dot_q6_K_q8_K_avx2(int64_t n, const orc_block_q6_K * x, const orc_block_q8_K * y);
/* Q5_0 / Q5_1 (arch/x86/quants.c:846-924, 926-1010): Q4_0 / Q4_1 with a fifth bit from qh.  IQ4_NL / MXFP4 (:3632-3714, 760-844): 16-entry int8 codebooks, TWO
 * 8-lane accumulators (even / odd blocks) added before the horizontal sum, a last odd block in scalar code; with >= 2 activation columns Q5_0 and IQ4_NL go through
 * tinyBLAS_Q0_AVX (llamafile/sgemm.cpp:1346-1790, dispatch :3984-4013): ONE accumulator per output, blocks in order (`chain` = 1 selects it).
 * Q2_K / Q3_K (:1278-1354, 1470-1580): per super-block acc[A] = fma(d, sumi[A], acc[A]); Q2_K first folds the mins: acc[A] = fma(dmin, m[2A] S[2A] + m[2A+1] S[2A+1], acc[A]) */
float orc_vec_dot_q5_0_q8_0_avx2(int64_t n, const orc_block_q5_0 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q5_1_q8_1_avx2(int64_t n, const orc_block_q5_1 * x, const orc_block_q8_1 * y);
float orc_vec_dot_iq4_nl_q8_0_avx2(int64_t n, const orc_block_iq4_nl * x, const orc_block_q8_0 * y, int chain);
/* IQ4_XS (arch/x86/quants.c:3716-3764): the K-quants' 8 lanes -- per super-block sumi[A] = sum over the eight 32-weight sub-blocks of (ls - 32) * (4-element codebook dot),
 * ONE fma(d_x * d_y, (float) sumi[A], acc[A]) per super-block; every column count goes through this dot product (no tinyBLAS case) */
float orc_vec_dot_iq4_xs_q8_K_avx2(int64_t n, const orc_block_iq4_xs * x, const orc_block_q8_K * y);
/* TQ1_0 / TQ2_0 (arch/x86/quants.c:1080-1210, 1212-1270): the 8 lanes again -- lane L takes elements 4L..4L+3 of every 32-element chunk; per super-block
 * sumi[L] = sum of trit * activation over its 32 elements - (bsums[2L] + bsums[2L+1]) (the "- 1" of every weight, taken from the activation's block sums lane-wise), then
 * sumf[L] = (float) sumi[L] * (d_y * d_x) + sumf[L] -- a multiply and an add, two roundings: the reference build does not contract `_mm256_add_ps(_mm256_mul_ps(..), sumf)` (pinned) */
/* ggml_vec_dot_iq2_xxs / iq2_xs / iq2_s / iq3_xxs / iq3_s _q8_K, AVX2 (arch/x86/quants.c:2372-3300): the 8 lanes -- lane L takes elements 4L..4L+3 of every 32-element
 * sub-block; sumi[L] = sum over sub-blocks of (odd integer scale of the lane's 16) * (signed codebook magnitudes . activation), ONE fma(d_x * d_y, (float) sumi[L], acc[L]) per
 * super-block, result = c * hsum_float_8(acc), c = 1/8 (IQ2), 1/4 (IQ3_XXS), 1 (IQ3_S) */
float orc_vec_dot_iq_grid_q8_K_avx2(int type, int64_t n, const void * x, const orc_block_q8_K * y);
/* ggml_vec_dot_iq1_s_q8_K / iq1_m (arch/x86/quants.c:3306-3362, 3425-3535): values -1 / 0 / 1 from iq1s_grid plus a per-group delta of +-1/8.  IQ1_S: the 8 lanes fold
 * fma(d, (float) sumi[L], acc[L]); the delta part is ONE scalar chain accum1 += d * sumi1 per super-block (sumi1 from the activation's bsums); result hsum(acc) + 0.125 accum1.
 * IQ1_M: two sets of 8 lanes (grid part, delta part), both fma per super-block; result hsum(acc1) + 0.125 hsum(acc2); d = fp16 assembled from the scales' top nibbles */
float orc_vec_dot_iq1_s_q8_K_avx2(int64_t n, const orc_block_iq1_s * x, const orc_block_q8_K * y);
float orc_vec_dot_iq1_m_q8_K_avx2(int64_t n, const orc_block_iq1_m * x, const orc_block_q8_K * y);
void orc_dequantize_row_iq1_s(const orc_block_iq1_s * x, float * y, int64_t k);    /* ggml-quants.c:2464-2486 */
void orc_dequantize_row_iq1_m(const orc_block_iq1_m * x, float * y, int64_t k);    /* ggml-quants.c:2488-2536 */
float orc_vec_dot_tq1_0_q8_K_avx2(int64_t n, const orc_block_tq1_0 * x, const orc_block_q8_K * y);
float orc_vec_dot_tq2_0_q8_K_avx2(int64_t n, const orc_block_tq2_0 * x, const orc_block_q8_K * y);
float orc_vec_dot_mxfp4_q8_0_avx2(int64_t n, const orc_block_mxfp4 * x, const orc_block_q8_0 * y);
float orc_vec_dot_q2_K_q8_K_avx2(int64_t n, const orc_block_q2_K * x, const orc_block_q8_K * y);
float orc_vec_dot_q3_K_q8_K_avx2(int64_t n, const orc_block_q3_K * x, const orc_block_q8_K * y);
float orc_vec_dot_f16_avx2(int64_t n, const uint16_t * x, const uint16_t * y);
float orc_vec_dot_f32_avx2(int64_t n, const float * x, const float * y);

/* ---- ops (dst written through its strides; all return 0 on success, <0 on bad arguments) ---- */
/* ggml_compute_forward_mul_mat (ggml-cpu/ggml-cpu.c:1229-1421): quantizes src1 rows to the
 * weight type's vec_dot_type (Q8_0 for Q4_0/Q8_0, Q8_1 for Q4_1, Q8_K for Q4_K, F16 for F16), then vec_dot. */
int orc_mul_mat(const orc_tensor * src0, const orc_tensor * src1, orc_tensor * dst);
/* ggml_compute_forward_mul_mat_id (ggml-cpu/ggml-cpu.c:1432-1678) */
int orc_mul_mat_id(const orc_tensor * as, const orc_tensor * b, const orc_tensor * ids, orc_tensor * dst);
/* ggml_compute_forward_rms_norm_f32 (ggml-cpu/ops.cpp:3710-3759): double accumulation */
int orc_rms_norm(const orc_tensor * src, orc_tensor * dst, float eps);

typedef struct orc_rope_params {         /* op_params of GGML_OP_ROPE (ggml.c ggml_rope_impl) */
    int32_t n_dims, mode, n_ctx_orig;    /* mode: 0 = NORMAL (pairs i,i+1), 2 = NEOX (i, i+n/2) */
    float   freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
} orc_rope_params;
/* ggml_compute_forward_rope_flt<float> (ggml-cpu/ops.cpp:5589-5865) */
int orc_rope(const orc_tensor * src, const int32_t * pos, const float * freq_factors, orc_tensor * dst,
             const orc_rope_params * p);
/* ggml_compute_forward_soft_max_f32 (ggml-cpu/ops.cpp:5225-5335; ggml_v_expf vec.h:1230-1267) */
int orc_soft_max(const orc_tensor * src, const orc_tensor * mask, orc_tensor * dst, float scale, float max_bias);
/* ggml_compute_forward_flash_attn_ext_f16_one_chunk (ggml-cpu/ops.cpp:8114-8344) -- the per-query online soft-max the reference
 * takes with one thread for N < 64 queries and n_kv < 512 (and always for a quantized K/V cache); its tiled (:8346-8640) and
 * split-KV (:8642-8770) variants differ from it in summation order only, so they are compared with a tolerance.
 * q F32 [D, N, H, B]; k, v F16 | Q8_0 [D, n_kv, Hkv, B]; mask NULL | F16 [n_kv, >= N, 1|H, 1|B]; dst F32 [D, H, N, B].
 * Q is converted to K's vec_dot_type (fp16 / quantize_row_q8_0), the dot is the AVX2-order vec_dot, V accumulates in fp16
 * (F16 V: ggml_vec_mad_f16, vec.h:456-) or fp32 (dequantized Q8_0 V: ggml_vec_mad_f32). */
int orc_flash_attn_ext(const orc_tensor * q, const orc_tensor * k, const orc_tensor * v, const orc_tensor * mask, orc_tensor * dst, float scale);
/* ggml_compute_forward_diag_mask_f32 (ggml-cpu/ops.cpp:5137-5185) value = -INF */
int orc_diag_mask_inf(const orc_tensor * src, orc_tensor * dst, int n_past);
/* ggml_compute_forward_scale (ops.cpp:4426-) y = x*s + b */
int orc_scale(const orc_tensor * src, orc_tensor * dst, float s, float b);
/* unary SILU (vec.cpp:396-431, vec.h:1270-1278): AVX2 polynomial for the first n&~7, expf tail */
int orc_silu(const orc_tensor * src, orc_tensor * dst);
/* binary-ops.cpp add / mul with broadcast of src1 over src0 */
int orc_add(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst);
int orc_mul(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst);
int orc_div(const orc_tensor * a, const orc_tensor * b, orc_tensor * dst);     /* vec.h:104: IEEE division */
/* the router of a sparse-MoE block (GenericSparseMLP::forward, src/layers.cpp:3755-3815) */
int orc_sum_rows(const orc_tensor * src, orc_tensor * dst);                     /* ops.cpp:1451-1482, double accumulator (vec.h:1510-1520) */
int orc_top_k(const orc_tensor * src, orc_tensor * dst);                        /* ops.cpp:8057-8094: descending, first two swapped */
/* ggml_compute_forward_set_rows_f32 (ops.cpp:4892-4940): dst rows (F16|F32) <- src rows (F32) at idx (I32|I64) */
int orc_set_rows(const orc_tensor * src, const orc_tensor * idx, orc_tensor * dst);
/* ggml_compute_forward_dup / cpy (ops.cpp:47-330,526): same #elements, F32->F32|F16, F16->F16|F32, any strides */
int orc_cpy(const orc_tensor * src, orc_tensor * dst);
/* ggml_compute_forward_get_rows (ops.cpp:4653-4700,4820): dst F32 rows <- dequant(src rows[idx]) */
int orc_get_rows(const orc_tensor * src, const orc_tensor * idx, orc_tensor * dst);

/* scalar helpers exposed for KATs */
float orc_expf_avx2(float x);   /* one lane of ggml_v_expf (vec.h:1230-1267), bit-exact */
float orc_silu_avx2(float x);   /* one lane of ggml_v_silu */

/* ---- whole-model restatement: Llama-3 / Qwen2 style decoder ---- */
typedef struct orc_llama_config {
    int32_t n_layer, hidden, n_head, n_kv_head, head_dim, ffn, vocab, max_len;
    int32_t rope_mode;          /* 0 interleaved (Llama-3, models/llama.h), 2 NEOX (Qwen2) */
    float   rope_theta, rms_eps;
    int32_t qkv_bias;           /* Qwen2 */
} orc_llama_config;

typedef struct orc_weight { int32_t type; const void * data; } orc_weight;  /* row-major [out][in] blocks */

typedef struct orc_llama_layer {
    const float * attn_norm, * ffn_norm;
    orc_weight wq, wk, wv, wo, wgate, wup, wdown;
    const float * bq, * bk, * bv;
} orc_llama_layer;

typedef struct orc_llama_model {
    orc_llama_config cfg;
    orc_weight tok_embd, lm_head;
    const float * out_norm;
    const orc_llama_layer * layers;
    /* KV cache, F16: k [layer][max_len][kvH*hd], v (eager layout, layers.cpp:3082-3093) [layer][kvH*hd][max_len] */
    uint16_t * k_cache, * v_cache;
} orc_llama_model;

/* run qlen tokens at positions n_past..n_past+qlen-1; logits[vocab] for the LAST token
 * (LMFinalSteps, src/models.cpp:1736-1784).  returns 0 or <0. */
int orc_llama_forward(orc_llama_model * m, const int32_t * tokens, int qlen, int n_past, float * logits);

#ifdef __cplusplus
}
#endif
#endif
