// dequant.h -- one element of a quantized row as dequantize_row_* produces it (ggml-quants.c:307-325 Q4_0, 327-345 Q4_1, 401-414 Q8_0,
// 1352-1373 Q4_K with get_scale_min_k4 :703-711): shared by GET_ROWS (ops.hip) and the dense fp16 prefill path (dense_f16.hip)
#pragma once
#include "common.h"
#include "iq_grids.h"

// the grid formats: for the 32-element sub-block ib of a block and the 8-group l (elements 8 l .. 8 l + 7 of it), the FOUR signed codebook magnitudes of its half
// `half` (int8 x 4), and the sub-block's scale nibble(s).  Shared by GET_ROWS (dequant_elem) and the mat-mul (gemv_kq.hip).  Layouts: ggml-common.h:346-390.
__device__ __forceinline__ uint32_t iq_grid_w4(int type, const char * blk, int ib, int l, int half, int & ls_lo, int & ls_hi) {
    const uint8_t * b8 = (const uint8_t *) blk;
    auto u16 = [&](int o) { return (uint32_t) b8[o] | ((uint32_t) b8[o + 1] << 8); };
    auto u32 = [&](int o) { return u16(o) | (u16(o + 2) << 16); };
    uint32_t mag, s8;
    if (type == CLLM_TYPE_IQ2_XXS) {
        const uint32_t a0 = u32(2 + 8 * ib), a1 = u32(2 + 8 * ib + 4);
        ls_lo = ls_hi = (int)(a1 >> 28);
        mag = iq2_code_bytes4(IQ2XXS_CODE[(a0 >> (8 * l)) & 0xff], 4 * half); s8 = iq_ksigns((a1 >> (7 * l)) & 127);
    } else if (type == CLLM_TYPE_IQ2_XS) {
        const uint32_t q = u16(2 + 2 * (4 * ib + l)), sc = b8[66 + ib];
        ls_lo = sc & 15; ls_hi = sc >> 4;
        mag = iq2_code_bytes4(IQ2XS_CODE[q & 511], 4 * half); s8 = iq_ksigns(q >> 9);
    } else if (type == CLLM_TYPE_IQ2_S) {
        const uint32_t sc = b8[74 + ib], idx = b8[2 + 4 * ib + l] | (((uint32_t) b8[66 + ib] << (8 - 2 * l)) & 0x300);
        ls_lo = sc & 15; ls_hi = sc >> 4;
        mag = iq2_code_bytes4(IQ2S_CODE[idx], 4 * half); s8 = b8[2 + 32 + 4 * ib + l];
    } else if (type == CLLM_TYPE_IQ3_XXS) {
        const uint32_t a = u32(2 + 64 + 4 * ib);
        ls_lo = ls_hi = (int)(a >> 28);
        mag = iq3xxs_code_bytes4(IQ3XXS_CODE[b8[2 + 8 * ib + 2 * l + half]]); s8 = iq_ksigns((a >> (7 * l)) & 127);
    } else {
        const uint32_t idx = b8[2 + 8 * ib + 2 * l + half] | (((uint32_t) b8[66 + ib] << (8 - 2 * l - half)) & 256);
        ls_lo = ls_hi = (b8[106 + ib / 2] >> (4 * (ib & 1))) & 15;
        mag = iq3s_code_bytes4(IQ3S_CODE[idx]); s8 = b8[74 + 4 * ib + l];
    }
    return iq_apply_signs4(mag, (s8 >> (4 * half)) & 15u);
}
// IQ1_S / IQ1_M (ggml-common.h:392-412): for sub-block ib and 8-group l the four values (-1 / 0 / 1, int8 x 4) of its half, the odd scale of the group's 16 and whether the
// group's delta is -1/8.  IQ1_S: d | qs[32] | qh[8] (u16: 4 x 3 index bits, 3-bit scale, delta sign); IQ1_M: qs[32] | qh[16] (nibbles: 3 index bits + delta sign) | scales[8]
__device__ __forceinline__ uint32_t iq1_w4(int type, const char * blk, int ib, int l, int half, int & ls, bool & neg) {
    const uint8_t * b8 = (const uint8_t *) blk;
    uint32_t idx;
    if (type == CLLM_TYPE_IQ1_S) {
        const uint32_t qh = (uint32_t) b8[34 + 2 * ib] | ((uint32_t) b8[35 + 2 * ib] << 8);
        idx = b8[2 + 4 * ib + l] | (((qh >> (3 * l)) & 7u) << 8);
        ls = 2 * (int)((qh >> 12) & 7u) + 1; neg = (qh & 0x8000u) != 0;
    } else {
        const uint32_t h = b8[32 + 2 * ib + (l >> 1)], sc = (uint32_t) b8[48 + 2 * (ib / 2)] | ((uint32_t) b8[49 + 2 * (ib / 2)] << 8);
        idx = b8[4 * ib + l] | (((l & 1) ? (h << 4) : (h << 8)) & 0x700u);
        ls = 2 * (int)((sc >> (6 * (ib % 2) + 3 * (l >> 1))) & 7u) + 1; neg = (h & ((l & 1) ? 0x80u : 0x08u)) != 0;
    }
    return iq1_code_bytes4(IQ1S_CODE[idx], 4 * half);
}
__device__ __forceinline__ float iq1m_d(const char * blk) {          // the fp16 scale in the top nibbles of the four 16-bit scale words
    const uint8_t * b8 = (const uint8_t *) blk + 48;
    return h2f((uint16_t)((b8[1] >> 4) | (b8[3] & 0xf0) | ((uint32_t)(b8[5] & 0xf0) << 4) | ((uint32_t)(b8[7] & 0xf0) << 8)));
}
__device__ __forceinline__ float dequant_elem(int type, const char * row, int64_t i) {
    switch (type) {
        case CLLM_TYPE_F32: return ((const float *) row)[i];
        case CLLM_TYPE_F16: return h2f(((const uint16_t *) row)[i]);
        case CLLM_TYPE_Q8_0: { const block_q8_0 * b = (const block_q8_0 *) row + i / 32; return (float) b->qs[i % 32] * h2f(b->d); }
        case CLLM_TYPE_Q4_0: {
            const block_q4_0 * b = (const block_q4_0 *) row + i / 32; const int j = (int)(i % 32);
            const int q = j < 16 ? (b->qs[j] & 0xF) : (b->qs[j - 16] >> 4);
            return (float)(q - 8) * h2f(b->d);
        }
        case CLLM_TYPE_Q4_1: {                     // dequantize_row_q4_1 (ggml-quants.c:327-345): nib * d + m
            const block_q4_1 * b = (const block_q4_1 *) row + i / 32; const int j = (int)(i % 32);
            const int q = j < 16 ? (b->qs[j] & 0xF) : (b->qs[j - 16] >> 4);
            return (float) q * h2f(b->d) + h2f(b->m);
        }
        case CLLM_TYPE_Q4_K: {
            const block_q4_K * b = (const block_q4_K *) row + i / 256; const int e = (int)(i % 256);
            const int s = e / 32, l = e % 32;
            int sc, m;
            if (s < 4) { sc = b->scales[s] & 63; m = b->scales[s + 4] & 63; }
            else { sc = (b->scales[s + 4] & 0xF) | ((b->scales[s - 4] >> 6) << 4); m = (b->scales[s + 4] >> 4) | ((b->scales[s] >> 6) << 4); }
            const uint8_t qb = b->qs[(s >> 1) * 32 + l];
            const int q = (s & 1) ? (qb >> 4) : (qb & 0xF);
            const float d1 = h2f(b->d) * (float) sc, m1 = h2f(b->dmin) * (float) m;     // d*sc and min*m rounded first (ggml-quants.c:1364-1367)
            return d1 * (float) q - m1;
        }
        case CLLM_TYPE_Q5_K: {                     // dequantize_row_q5_K (ggml-quants.c:1554-1579)
            const block_q5_K * b = (const block_q5_K *) row + i / 256; const int e = (int)(i % 256);
            const int s = e / 32, l = e % 32;
            int sc, m;
            if (s < 4) { sc = b->scales[s] & 63; m = b->scales[s + 4] & 63; }
            else { sc = (b->scales[s + 4] & 0xF) | ((b->scales[s - 4] >> 6) << 4); m = (b->scales[s + 4] >> 4) | ((b->scales[s] >> 6) << 4); }
            const uint8_t qb = b->qs[(s >> 1) * 32 + l];
            const int q = ((s & 1) ? (qb >> 4) : (qb & 0xF)) + (((b->qh[l] >> s) & 1) ? 16 : 0);
            const float d1 = h2f(b->d) * (float) sc, m1 = h2f(b->dmin) * (float) m;
            return d1 * (float) q - m1;
        }
        case CLLM_TYPE_Q6_K: {                     // dequantize_row_q6_K (ggml-quants.c:1762-1791): (d * sc) * q, q = 6 bits - 32
            const block_q6_K * b = (const block_q6_K *) row + i / 256; const int e = (int)(i % 256);
            const int j = e / 128, g = (e % 128) / 32, l = e % 32;
            const uint8_t lb = b->ql[64 * j + 32 * (g & 1) + l];
            const int lo = g < 2 ? (lb & 0xF) : (lb >> 4);
            const int q = (int)(int8_t)(lo | (((b->qh[32 * j + l] >> (2 * g)) & 3) << 4)) - 32;
            const float ds = h2f(b->d) * (float) b->scales[8 * j + 2 * g + l / 16];
            return ds * (float) q;
        }
        case CLLM_TYPE_Q5_0: case CLLM_TYPE_Q5_1: {   // dequantize_row_q5_0 / _q5_1 (ggml-quants.c:348-399)
            const bool q51 = type == CLLM_TYPE_Q5_1;
            const uint8_t * b = (const uint8_t *) row + (i / 32) * (q51 ? 24 : 22); const int j = (int)(i % 32);
            const uint8_t * qh = b + (q51 ? 4 : 2), * qs = qh + 4;
            const uint32_t h = (uint32_t) qh[0] | ((uint32_t) qh[1] << 8) | ((uint32_t) qh[2] << 16) | ((uint32_t) qh[3] << 24);
            const int q = (j < 16 ? (qs[j] & 0xF) : (qs[j - 16] >> 4)) | (int)(((h >> j) & 1u) << 4);
            const float d = h2f((uint16_t)(b[0] | (b[1] << 8)));
            return q51 ? (float) q * d + h2f((uint16_t)(b[2] | (b[3] << 8))) : (float)(q - 16) * d;
        }
        case CLLM_TYPE_IQ4_NL: case CLLM_TYPE_MXFP4: {   // dequantize_row_iq4_nl / _mxfp4 (ggml-quants.c:2512-2528, 417-436)
            const bool mx = type == CLLM_TYPE_MXFP4;
            const uint8_t * b = (const uint8_t *) row + (i / 32) * (mx ? 17 : 18); const int j = (int)(i % 32);
            const uint8_t * qs = b + (mx ? 1 : 2);
            const int nib = j < 16 ? (qs[j] & 0xF) : (qs[j - 16] >> 4);
            const uint64_t tab = mx ? (nib < 8 ? 0x0c08060403020100ull : 0xf4f8fafcfdfeff00ull) : (nib < 8 ? 0xf6eaddcfbfad9881ull : 0x7159453526190d01ull);
            const float v = (float)(int8_t)((tab >> (8 * (nib & 7))) & 0xff);
            if (mx) return v * __uint_as_float(b[0] < 2 ? 0x00200000u << b[0] : (uint32_t)(b[0] - 1) << 23);
            return h2f((uint16_t)(b[0] | (b[1] << 8))) * v;
        }
        case CLLM_TYPE_IQ4_XS: {                   // dequantize_row_iq4_xs (ggml-quants.c:2530-2551): (d * (ls - 32)) * kvalues_iq4nl[nib]
            const block_iq4_xs * b = (const block_iq4_xs *) row + i / 256; const int e = (int)(i % 256), ib = e / 32, j = e % 32;
            const int ls = ((b->scales_l[ib / 2] >> (4 * (ib % 2))) & 0xf) | (((b->scales_h >> (2 * ib)) & 3) << 4);
            const float dl = h2f(b->d) * (float)(ls - 32);
            const uint8_t qb = b->qs[16 * ib + (j & 15)];
            const int nib = j < 16 ? (qb & 0xF) : (qb >> 4);
            const uint64_t tab = nib < 8 ? 0xf6eaddcfbfad9881ull : 0x7159453526190d01ull;
            return dl * (float)(int8_t)((tab >> (8 * (nib & 7))) & 0xff);
        }
        case CLLM_TYPE_IQ1_S: case CLLM_TYPE_IQ1_M: {  // dequantize_row_iq1_s / _iq1_m (ggml-quants.c:2464-2536): y = dl * (value + delta), dl = d * (2 s + 1), delta = +-0.125f
            const char * blk = row + (i / 256) * (type == CLLM_TYPE_IQ1_S ? 50 : 56); const int e = (int)(i % 256), ib = e / 32, l = (e % 32) / 8, half = (e % 8) / 4, k = e % 4;
            int ls; bool neg;
            const uint32_t w4 = iq1_w4(type, blk, ib, l, half, ls, neg);
            const float d = type == CLLM_TYPE_IQ1_S ? h2f(*(const uint16_t *) blk) : iq1m_d(blk);
            return (d * (float) ls) * ((float)(int8_t)((w4 >> (8 * k)) & 0xff) + (neg ? -0.125f : 0.125f));
        }
        case CLLM_TYPE_IQ2_XXS: case CLLM_TYPE_IQ2_XS: case CLLM_TYPE_IQ2_S: case CLLM_TYPE_IQ3_XXS: case CLLM_TYPE_IQ3_S: {
            // dequantize_row_iq2_xxs / _iq2_xs / _iq2_s / _iq3_xxs / _iq3_s (ggml-quants.c:2275-2460): y = db * magnitude * (+-1.f), db = d * (0.5f + ls) * 0.25f (IQ2), * 0.5f
            // (IQ3_XXS), d * (1 + 2 ls) (IQ3_S) -- in these operation orders
            const int bs = type == CLLM_TYPE_IQ2_XXS ? 66 : type == CLLM_TYPE_IQ2_XS ? 74 : type == CLLM_TYPE_IQ2_S ? 82 : type == CLLM_TYPE_IQ3_XXS ? 98 : 110;
            const char * blk = row + (i / 256) * bs; const int e = (int)(i % 256), ib = e / 32, l = (e % 32) / 8, half = (e % 8) / 4, k = e % 4;
            int ls_lo, ls_hi;
            const uint32_t w4 = iq_grid_w4(type, blk, ib, l, half, ls_lo, ls_hi);
            const int w = (int)(int8_t)((w4 >> (8 * k)) & 0xff), ls = (e % 32) < 16 ? ls_lo : ls_hi;
            const float d = h2f(*(const uint16_t *) blk);
            const float db = type == CLLM_TYPE_IQ3_S ? d * (float)(1 + 2 * ls) : type == CLLM_TYPE_IQ3_XXS ? d * (0.5f + (float) ls) * 0.5f : d * (0.5f + (float) ls) * 0.25f;
            return db * (float)(w < 0 ? -w : w) * (w < 0 ? -1.0f : 1.0f);
        }
        case CLLM_TYPE_TQ1_0: {                    // dequantize_row_tq1_0 (ggml-quants.c:2215-2252): trit = ((byte * 3^n mod 256) * 3) >> 8, planes of 32 / 16 / 4 bytes
            const block_tq1_0 * b = (const block_tq1_0 *) row + i / 256; const int e = (int)(i % 256);
            int byte, n;
            if (e < 160)      { byte = b->qs[e % 32];              n = e / 32; }
            else if (e < 240) { byte = b->qs[32 + (e - 160) % 16]; n = (e - 160) / 16; }
            else              { byte = b->qh[(e - 240) % 4];       n = (e - 240) / 4; }
            const int p3 = n == 0 ? 1 : n == 1 ? 3 : n == 2 ? 9 : n == 3 ? 27 : 81;
            return (float)((int)((((unsigned)(byte * p3) & 0xffu) * 3u) >> 8) - 1) * h2f(b->d);
        }
        case CLLM_TYPE_TQ2_0: {                    // dequantize_row_tq2_0 (ggml-quants.c:2254-2271): (q - 1) * d, 2 bits per weight, planes of 32 bytes
            const block_tq2_0 * b = (const block_tq2_0 *) row + i / 256; const int e = (int)(i % 256);
            const int q = (b->qs[32 * (e / 128) + e % 32] >> (2 * ((e % 128) / 32))) & 3;
            return (float)(q - 1) * h2f(b->d);
        }
        case CLLM_TYPE_Q2_K: {                     // dequantize_row_q2_K (ggml-quants.c:784-815): (d * sc) * q - (dmin * m)
            const block_q2_K * b = (const block_q2_K *) row + i / 256; const int e = (int)(i % 256);
            const int n = e / 128, j = (e % 128) / 32, hh = (e % 32) / 16, l = e % 16;
            const uint8_t sc = b->scales[8 * n + 2 * j + hh];
            const float dl = h2f(b->d) * (float)(sc & 0xF), ml = h2f(b->dmin) * (float)(sc >> 4);
            return dl * (float)((b->qs[32 * n + 16 * hh + l] >> (2 * j)) & 3) - ml;
        }
        case CLLM_TYPE_Q3_K: {                     // dequantize_row_q3_K (ggml-quants.c:1128-1176)
            const block_q3_K * b = (const block_q3_K *) row + i / 256; const int e = (int)(i % 256);
            const int n = e / 128, j = (e % 128) / 32, hh = (e % 32) / 16, l = e % 16, is = 8 * n + 2 * j + hh, bb = 16 * hh + l;
            const uint8_t * s = b->scales;      // 6-bit scale `is`: low 4 bits in bytes 0..7 (nibbles), high 2 bits in bytes 8..11
            const int lo = is < 8 ? (s[is] & 0xF) : (s[is - 8] >> 4), hi = (s[8 + (is & 3)] >> (2 * (is >> 2))) & 3;
            const float dl = h2f(b->d) * (float)((lo | (hi << 4)) - 32);
            const int q = (int)((b->qs[32 * n + bb] >> (2 * j)) & 3) - (((b->hmask[bb] >> (4 * n + j)) & 1) ? 0 : 4);
            return dl * (float) q;
        }
    }
    return 0.0f;
}
