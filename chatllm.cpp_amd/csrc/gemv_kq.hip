// gemv_kq.hip -- MUL_MAT for the COVERAGE weight types: every format a stock chatllm.cpp / GGML model file may carry besides the tuned ones (Q4_K_M / Q5_K_M mixes
// keep tensors in Q5_K and Q6_K; older and third-party files use Q5_0 / Q5_1 / Q2_K / Q3_K / IQ4_NL; MXFP4), for any number of columns, BIT-IDENTICAL to the
// reference's x86 AVX2 dot products (arch/x86/quants.c):
//   q5_K :1916-2030   8 lane accumulators acc[A] = fma(y.d x.d, (float) sumi[A], acc[A]) per super-block in order, + ONE scalar chain summs += (-y.d x.dmin) * sum m S
//   q6_K :2130-2225   the same lanes, int8 scale per 16 elements, no mins
//   q2_K :1278-1354   acc[A] = fma(dmin, m[2A] S16[2A] + m[2A+1] S16[2A+1], acc[A]) then acc[A] = fma(d, sumi[A], acc[A]) (S16: activation sums per 16)
//   q3_K :1470-1580   2 bits + a high-bit mask (value - 4 where the bit is clear), 6-bit scales - 32
//   q5_0 :846-884, q5_1 :926-968    Q4_0's / Q4_1's chains with a fifth bit per weight
//   iq2_xxs :2372-, iq2_xs :2490-, iq2_s :2787-, iq3_xxs :2972-, iq3_s :3096-   codebook ("grid") formats: signed magnitudes from iq_grids.h, an odd integer scale 2 ls + 1 per 16 or 32,
//                     one fma per super-block and lane, the result scaled by 1/8 (IQ2), 1/4 (IQ3_XXS), 1 (IQ3_S) AFTER the horizontal sum
//   iq1_s :3306-3362, iq1_m :3425-3535   values -1 / 0 / 1 from the IQ1 codebook + a delta of +-1/8 per group: IQ1_S folds the delta part as ONE scalar chain accum1 += d * sumi1
//                     (two roundings; sumi1 from the activation's block sums), IQ1_M as a second set of 8 lanes; result hsum(acc) + 0.125 (accum1 | hsum(acc2))
//   tq1_0 :1080-1210, tq2_0 :1212-1270   ternary weights; sumi[A] - (bsums[2A] + bsums[2A+1]), then (float) sumi * d + acc as TWO roundings (the build keeps the multiply and the add)
//   iq4_xs :3716-3764   256-weight super-blocks of IQ4_NL codes with a 6-bit scale - 32 per 32: the K-quants' single fma per super-block and lane
//   iq4_nl :3632-3714, mxfp4 :760-844   int8 codebooks; even blocks in one 8-lane accumulator, odd ones in a second, added before the horizontal sum, an unpaired
//                     last block in scalar code; with >= 2 activation columns IQ4_NL (and Q5_0) take tinyBLAS_Q0_AVX (llamafile/sgemm.cpp:1346-1790): ONE chain
//   result = hsum_float_8(acc) [+ summs]
// AVX lane A = dword A of every 32-byte chunk.  The simple, obviously-exact mapping: 8 GPU lanes per weight row, lane A computes sumi[A] of every
// block itself and carries acc[A] in a register, so the serial fp32 chain needs no cross-lane traffic at all; the three adds of
// hsum_float_8 are lane exchanges at the end.  Activations: the act rows of quantize.hip (common.h layout: Q8_K, Q8_0 or Q8_1 kind).  Blocks that are only
// 2-byte (1-byte: MXFP4) aligned are read in 16-bit (8-bit) pieces.  This is the coverage path (it streams at a fraction of the Q4_K kernels' rate), not a tuned one.
#include "common.h"
#include "dequant.h"
#include "q4k.h"


struct kq_args {
    const char * W; int64_t nb01; int N, nblk;
    const char * act; int64_t act_stride; int off_d, off_s;
    float * dst; int64_t ldd; int ncols;
    // MUL_MAT_ID (ids != nullptr): blockIdx.y = token * n_used + slot; the expert comes from device memory, one column per launch slice
    const int32_t * ids; int64_t ids_nb1 /* ints */, nb02, dst_tok /* floats */; int n_used, ne11;
};
// expert / activation row / destination of this grid slice (MUL_MAT_ID, ggml_compute_forward_mul_mat_id ggml-cpu.c:1432-1678: dst[:, slot, tok] = as[:, :, ids[slot, tok]]^T . b[:, slot % ne11, tok])
__device__ __forceinline__ void kq_slice(const kq_args & a, const char * & W, const char * & act, float * & dst) {
    W = a.W; act = a.act; dst = a.dst;
    if (a.ids) {
        const int pair = blockIdx.y, tok = pair / a.n_used, slot = pair - tok * a.n_used;
        const int e = a.ids[(int64_t) tok * a.ids_nb1 + slot];
        W += (int64_t) e * a.nb02;
        act += (int64_t)(tok * a.ne11 + (a.ne11 == 1 ? 0 : slot)) * a.act_stride;
        dst += (int64_t) tok * a.dst_tok + (int64_t) slot * a.ldd;
    }
}

__device__ __forceinline__ uint32_t ld2(const char * p) { return (uint32_t) *(const uint16_t *) p | ((uint32_t) *(const uint16_t *)(p + 2) << 16); }

__device__ __forceinline__ uint32_t ld1x4(const char * p) { const uint8_t * q = (const uint8_t *) p; return (uint32_t) q[0] | ((uint32_t) q[1] << 8) | ((uint32_t) q[2] << 16) | ((uint32_t) q[3] << 24); }
// four 4-bit indices (one per byte of idx) -> four bytes of a 16-entry int8 table held as four dwords
__device__ __forceinline__ uint32_t lut16(uint32_t idx, uint32_t t0, uint32_t t1, uint32_t t2, uint32_t t3) {
    const uint32_t sel = idx & 0x07070707u;
    const uint32_t lo = __builtin_amdgcn_perm(t1, t0, sel), hi = __builtin_amdgcn_perm(t3, t2, sel);
    const uint32_t m = ((idx >> 3) & 0x01010101u) * 0xffu;
    return (hi & m) | (lo & ~m);
}
template <int TYPE, int NC>
__global__ void __launch_bounds__(256) k_gemv_kq(const kq_args a) {
    const int tid = blockIdx.x * 256 + threadIdx.x, A = tid & 7;
    int row = tid >> 3;
    const bool live = row < a.N;                       // whole waves stay: the final lane exchanges read their neighbours
    if (!live) row = a.N - 1;
    const char * Wb, * actb; float * dstb;
    kq_slice(a, Wb, actb, dstb);
    const char * wr = Wb + (int64_t) row * a.nb01;
    float acc[NC], summs[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { acc[c] = 0.0f; summs[c] = 0.0f; }

    for (int b = 0; b < a.nblk; b++) {
        int8_t sc8[8]; uint32_t w[8];                  // the 8 (scale, int8 x 4) terms of this lane in this super-block, in activation order
        int aoff[8];
        float dw, dmin = 0.0f; int mn[8];
        if (TYPE == CLLM_TYPE_Q5_K) {
            const char * blk = wr + (int64_t) b * 176;
            const u32x4 h = *(const u32x4 *) blk;
            dw = h2f((uint16_t)(h.x & 0xffff)); dmin = h2f((uint16_t)(h.x >> 16));
            const uint32_t u0 = h.y & 0x3f3f3f3fu, u2 = h.z & 0x3f3f3f3fu;                          // get_scale_min_k4 (ggml-quants.c:703-711)
            const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
            const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
            const uint32_t qh = *(const uint32_t *)(blk + 16 + 4 * A);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const uint32_t q = *(const uint32_t *)(blk + 48 + 32 * c + 4 * A);
                w[2 * c]     = (q & 0x0f0f0f0fu)        | (((qh >> (2 * c))     & 0x01010101u) << 4);
                w[2 * c + 1] = ((q >> 4) & 0x0f0f0f0fu) | (((qh >> (2 * c + 1)) & 0x01010101u) << 4);
                aoff[2 * c] = 64 * c + 4 * A; aoff[2 * c + 1] = 64 * c + 32 + 4 * A;
            }
#pragma unroll
            for (int s = 0; s < 8; s++) {
                sc8[s] = (int8_t)(((s < 4 ? u0 : u1) >> (8 * (s & 3))) & 0xff);
                mn[s]  = (int)(((s < 4 ? u2 : u3) >> (8 * (s & 3))) & 0xff);
            }
        } else if (TYPE == CLLM_TYPE_Q2_K || TYPE == CLLM_TYPE_Q3_K) {
            // term t = 4 n + j: plane j (bits 2j) of the 32-byte vector of the n-th 128 weights, bytes 4A..4A+3, against activations 128 n + 32 j + 4A..; its scale is the one of
            // the 16-weight sub-block the bytes lie in: index 8 n + 2 j + (A >> 2)
            constexpr bool Q2 = TYPE == CLLM_TYPE_Q2_K;
            const char * blk = wr + (int64_t) b * (Q2 ? 84 : 110);
            dw = h2f(*(const uint16_t *)(blk + (Q2 ? 80 : 108)));
            int8_t s16[16];
            if (Q2) {
                dmin = h2f(*(const uint16_t *)(blk + 82));
#pragma unroll
                for (int k = 0; k < 4; k++) { const uint32_t v = ld2(blk + 4 * k);
#pragma unroll
                    for (int i = 0; i < 4; i++) s16[4 * k + i] = (int8_t)((v >> (8 * i)) & 0xff); }       // scale | min << 4
            } else {
                const uint32_t a0 = ld2(blk + 96), a1 = ld2(blk + 100), a2 = ld2(blk + 104);               // the sixteen 6-bit scales (ggml-quants.c:1141-1150), - 32
                const uint32_t u[4] = { (a0 & 0x0f0f0f0fu) | (((a2 >> 0) & 0x03030303u) << 4), (a1 & 0x0f0f0f0fu) | (((a2 >> 2) & 0x03030303u) << 4),
                                        ((a0 >> 4) & 0x0f0f0f0fu) | (((a2 >> 4) & 0x03030303u) << 4), ((a1 >> 4) & 0x0f0f0f0fu) | (((a2 >> 6) & 0x03030303u) << 4) };
#pragma unroll
                for (int k = 0; k < 4; k++)
#pragma unroll
                    for (int i = 0; i < 4; i++) s16[4 * k + i] = (int8_t)((int)((u[k] >> (8 * i)) & 0xff) - 32);
            }
            const uint32_t hm = Q2 ? 0u : ld2(blk + 4 * A);
#pragma unroll
            for (int n = 0; n < 2; n++) {
                const uint32_t q = ld2(blk + (Q2 ? 16 : 32) + 32 * n + 4 * A);
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int t = 4 * n + j;
                    uint32_t v = (q >> (2 * j)) & 0x03030303u;
                    if (!Q2) { v |= ((hm >> (4 * n + j)) & 0x01010101u) << 2; v = ((v | 0x80808080u) - 0x04040404u) ^ 0x80808080u; }      // per byte: (q | h << 2) - 4
                    w[t] = v;
                    const int8_t sc = (A >> 2) ? s16[8 * n + 2 * j + 1] : s16[8 * n + 2 * j];
                    sc8[t] = Q2 ? (int8_t)(sc & 0xF) : sc;
                    aoff[t] = 128 * n + 32 * j + 4 * A;
                }
            }
            if (Q2) { mn[0] = ((uint8_t) s16[2 * A]) >> 4; mn[1] = ((uint8_t) s16[2 * A + 1]) >> 4; }
        } else if (is_iq_grid_type(TYPE)) {
            // sub-block ib, lane A: the 8-group l = A >> 1, its half A & 1 -> four signed magnitudes; the scale of the lane's 16: 2 ls + 1
            constexpr int BS = TYPE == CLLM_TYPE_IQ2_XXS ? 66 : TYPE == CLLM_TYPE_IQ2_XS ? 74 : TYPE == CLLM_TYPE_IQ2_S ? 82 : TYPE == CLLM_TYPE_IQ3_XXS ? 98 : 110;
            const char * blk = wr + (int64_t) b * BS;
            dw = h2f(*(const uint16_t *) blk);
#pragma unroll
            for (int ib = 0; ib < 8; ib++) {
                int ls_lo, ls_hi;
                w[ib] = iq_grid_w4(TYPE, blk, ib, A >> 1, A & 1, ls_lo, ls_hi);
                sc8[ib] = (int8_t)(2 * ((A >> 2) ? ls_hi : ls_lo) + 1);
                aoff[ib] = 32 * ib + 4 * A;
            }
        } else if (TYPE == CLLM_TYPE_IQ1_S || TYPE == CLLM_TYPE_IQ1_M) {
            // sub-block ib, lane A: 8-group l = A >> 1, half A & 1 -> four values in {-1, 0, 1}; mn[ib] = the group's scale with the delta's sign (IQ1_M: per lane; IQ1_S: per sub-block)
            const char * blk = wr + (int64_t) b * (TYPE == CLLM_TYPE_IQ1_S ? 50 : 56);
            dw = TYPE == CLLM_TYPE_IQ1_S ? h2f(*(const uint16_t *) blk) : iq1m_d(blk);
#pragma unroll
            for (int ib = 0; ib < 8; ib++) {
                int ls; bool neg;
                w[ib] = iq1_w4(TYPE, blk, ib, A >> 1, A & 1, ls, neg);
                sc8[ib] = (int8_t) ls; mn[ib] = neg ? -ls : ls;
                aoff[ib] = 32 * ib + 4 * A;
            }
        } else if (TYPE == CLLM_TYPE_TQ2_0) {
            // plane l (bits 2l) of the j-th 32 bytes = elements 128 j + 32 l + (0..31): lane A takes bytes 4A..4A+3 of every plane; the weight is q - 1, the "- 1" comes off below
            const char * blk = wr + (int64_t) b * 66;
            dw = h2f(*(const uint16_t *)(blk + 64));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t q = ld2(blk + 32 * j + 4 * A);
#pragma unroll
                for (int l = 0; l < 4; l++) { w[4 * j + l] = (q >> (2 * l)) & 0x03030303u; sc8[4 * j + l] = 1; aoff[4 * j + l] = 128 * j + 32 * l + 4 * A; }
            }
        } else if (TYPE == CLLM_TYPE_TQ1_0) {
            // chunk c of 32 elements: c < 5: plane c of qs[0..31] (bytes 4A..); c = 5, 6: planes (0, 1), (2, 3) of qs[32..47] (the low half of the chunk from the even plane);
            // c = 7: plane 4 of qs[32..47], then the 16 elements of qh (plane A - 4 of its 4 bytes).  trit = ((byte * 3^n mod 256) * 3) >> 8 per byte, two 16-bit fields at a time
            const char * blk = wr + (int64_t) b * 54;
            dw = h2f(*(const uint16_t *)(blk + 52));
            const uint32_t B0 = ld2(blk + 4 * A), B1 = ld2(blk + 32 + 4 * (A & 3)), QH = ld2(blk + 48);
            auto trits = [](uint32_t bytes4, uint32_t p) -> uint32_t {
                uint32_t ev = ((bytes4 & 0x00ff00ffu) * p) & 0x00ff00ffu, od = (((bytes4 >> 8) & 0x00ff00ffu) * p) & 0x00ff00ffu;
                ev = ((ev * 3u) >> 8) & 0x00030003u; od = ((od * 3u) >> 8) & 0x00030003u;
                return ev | (od << 8);
            };
            const bool hi = (A >> 2) != 0;
            const uint32_t p3a = A == 4 ? 1u : A == 5 ? 3u : A == 6 ? 9u : 27u;
            w[0] = trits(B0, 1u); w[1] = trits(B0, 3u); w[2] = trits(B0, 9u); w[3] = trits(B0, 27u); w[4] = trits(B0, 81u);
            w[5] = trits(B1, hi ? 3u : 1u); w[6] = trits(B1, hi ? 27u : 9u);
            w[7] = hi ? trits(QH, p3a) : trits(B1, 81u);
#pragma unroll
            for (int c8 = 0; c8 < 8; c8++) { sc8[c8] = 1; aoff[c8] = 32 * c8 + 4 * A; }
        } else if (TYPE == CLLM_TYPE_IQ4_XS) {
            // sub-block ib (32 weights = 16 bytes: low nibbles are elements 0..15, high nibbles 16..31): lane A takes elements 4A..4A+3 = the low (A < 4) or high nibbles of
            // bytes 4 (A & 3) ..+3 through the IQ4_NL codebook; its scale is the 6-bit ls - 32 (scales_l nibble | two bits of scales_h)
            const char * blk = wr + (int64_t) b * 136;
            const uint32_t hd = *(const uint32_t *) blk, sl = *(const uint32_t *)(blk + 4);        // d | scales_h << 16, scales_l[4]
            dw = h2f((uint16_t)(hd & 0xffff));
            const uint32_t sh = hd >> 16;
#pragma unroll
            for (int ib = 0; ib < 8; ib++) {
                const uint32_t q4 = *(const uint32_t *)(blk + 8 + 16 * ib + 4 * (A & 3));
                w[ib] = lut16((q4 >> ((A >> 2) * 4)) & 0x0f0f0f0fu, 0xbfad9881u, 0xf6eaddcfu, 0x26190d01u, 0x71594535u);      // kvalues_iq4nl (ggml-common.h:1088-1090)
                sc8[ib] = (int8_t)((int)(((sl >> (4 * ib)) & 0xfu) | (((sh >> (2 * ib)) & 3u) << 4)) - 32);
                aoff[ib] = 32 * ib + 4 * A;
            }
        } else {
            const char * blk = wr + (int64_t) b * 210;
            dw = h2f(*(const uint16_t *)(blk + 208));
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const uint32_t l0 = ld2(blk + 64 * j + 4 * A), l1 = ld2(blk + 64 * j + 32 + 4 * A), qh = ld2(blk + 128 + 32 * j + 4 * A);
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const uint32_t lo = ((g & 1) ? l1 : l0) >> (g < 2 ? 0 : 4) & 0x0f0f0f0fu;
                    const uint32_t q6 = lo | (((qh >> (2 * g)) & 0x03030303u) << 4);
                    w[4 * j + g] = ((q6 | 0x80808080u) - 0x20202020u) ^ 0x80808080u;              // per byte: q - 32
                    const uint32_t sp = *(const uint16_t *)(blk + 192 + 8 * j + 2 * g);           // scales of the group's two 16-element halves
                    sc8[4 * j + g] = (int8_t)((A >> 2) ? sp >> 8 : sp & 0xff);
                    aoff[4 * j + g] = 128 * j + 32 * g + 4 * A;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int cc = c < a.ncols ? c : 0;
            const char * ar = actb + (int64_t) cc * a.act_stride;
            int sumi = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) sumi += (int) sc8[t] * dot4(w[t], *(const uint32_t *)(ar + b * 256 + aoff[t]), 0);
            const float yd = ((const float *)(ar + a.off_d))[b];
            if (TYPE == CLLM_TYPE_Q2_K) {                                      // the mins first: S16[k] = sum of the activation quants 16 k .. 16 k + 15
                int s16a = 0, s16b = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    s16a = dot4(0x01010101u, *(const uint32_t *)(ar + b * 256 + 32 * A + 4 * k), s16a);
                    s16b = dot4(0x01010101u, *(const uint32_t *)(ar + b * 256 + 32 * A + 16 + 4 * k), s16b);
                }
                acc[c] = __builtin_fmaf((-yd) * dmin, (float)(mn[0] * s16a + mn[1] * s16b), acc[c]);
            }
            if (TYPE == CLLM_TYPE_IQ1_S) {                                     // the delta part: sumi1 = sum over sub-blocks of (activation sum of the 32) * (+-scale); accum1 += d * sumi1 (two roundings)
                const int * ys = (const int *)(ar + a.off_s) + b * 8;
                int s1 = 0;
#pragma unroll
                for (int ib = 0; ib < 8; ib++) s1 += ys[ib] * mn[ib];
                summs[c] = summs[c] + (yd * dw) * (float) s1;
            }
            if (TYPE == CLLM_TYPE_IQ1_M) {                                     // the delta part on the same 8 lanes: sumi2[A] = sum of (+-scale) * (activation sum of the lane's 4)
                int s2 = 0;
#pragma unroll
                for (int t = 0; t < 8; t++) s2 += mn[t] * dot4(0x01010101u, *(const uint32_t *)(ar + b * 256 + aoff[t]), 0);
                summs[c] = __builtin_fmaf(yd * dw, (float) s2, summs[c]);
            }
            if (TYPE == CLLM_TYPE_TQ1_0 || TYPE == CLLM_TYPE_TQ2_0) {          // every weight is (value - 1): minus the activation's sum over this lane's 32 elements (bsums[2A] + bsums[2A+1])
                sumi -= ((const int *)(ar + a.off_s))[b * 8 + A];
                acc[c] = (float) sumi * (yd * dw) + acc[c];                    // a multiply and an add: two roundings, as the reference build has them
            } else
            acc[c] = __builtin_fmaf(yd * dw, (float) sumi, acc[c]);
            if (TYPE == CLLM_TYPE_Q5_K) {
                const int * ys = (const int *)(ar + a.off_s) + b * 8;
                int hs = 0;
#pragma unroll
                for (int s = 0; s < 8; s++) hs += mn[s] * ys[s];
                summs[c] = summs[c] + ((-yd) * dmin) * (float) hs;            // two roundings, as the reference build does it
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float v = acc[c];
        v = v + __int_as_float(lane_xor4_i(__float_as_int(v)));               // x[A] + x[A + 4]
        v = v + dpp_f<DPP_QUAD_XOR2>(v);                                      // (.. 0) + (.. 2), (.. 1) + (.. 3)
        v = v + dpp_f<DPP_QUAD_XOR1>(v);
        if (TYPE == CLLM_TYPE_Q5_K) v = v + summs[c];
        if (is_iq_grid_type(TYPE)) v = (TYPE == CLLM_TYPE_IQ3_S ? 1.0f : TYPE == CLLM_TYPE_IQ3_XXS ? 0.25f : 0.125f) * v;
        if (TYPE == CLLM_TYPE_IQ1_S) v = v + 0.125f * summs[c];              // hsum_float_8(accum) + IQ1S_DELTA * accum1 (every lane holds the same scalar chain)
        if (TYPE == CLLM_TYPE_IQ1_M) {                                        // hsum_float_8(accum1) + IQ1M_DELTA * hsum_float_8(accum2)
            float u = summs[c];
            u = u + __int_as_float(lane_xor4_i(__float_as_int(u))); u = u + dpp_f<DPP_QUAD_XOR2>(u); u = u + dpp_f<DPP_QUAD_XOR1>(u);
            v = v + 0.125f * u;
        }
        if (live && A == 0 && c < a.ncols) dstb[(int64_t) c * a.ldd + row] = v;
    }
}

// ---- the 32-weight block formats: Q5_0 / Q5_1 / IQ4_NL / MXFP4.  Lane A owns elements 4A..4A+3 of every block: the low (A < 4) or high nibbles of quant bytes
//      4 (A & 3) .. + 3.  CHAIN: one accumulator (Q5_0 / Q5_1 always; IQ4_NL with >= 2 columns); else even / odd blocks in two, the unpaired last block in scalar order ----
template <int TYPE, int NC, bool CHAIN>
__global__ void __launch_bounds__(256) k_gemv_b32(const kq_args a) {
    constexpr bool Q50 = TYPE == CLLM_TYPE_Q5_0, Q51 = TYPE == CLLM_TYPE_Q5_1, NL = TYPE == CLLM_TYPE_IQ4_NL, MX = TYPE == CLLM_TYPE_MXFP4;
    constexpr int BS = Q50 ? 22 : Q51 ? 24 : NL ? 18 : 17, QOFF = Q50 ? 6 : Q51 ? 8 : NL ? 2 : 1;
    const int tid = blockIdx.x * 256 + threadIdx.x, A = tid & 7;
    int row = tid >> 3;
    const bool live = row < a.N;
    if (!live) row = a.N - 1;
    const char * Wb, * actb; float * dstb;
    kq_slice(a, Wb, actb, dstb);
    const char * wr = Wb + (int64_t) row * a.nb01;
    float acc[NC], acc2[NC], summs[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) { acc[c] = 0.0f; acc2[c] = 0.0f; summs[c] = 0.0f; }
    const int npair = CHAIN ? a.nblk : (a.nblk & ~1);
    // kvalues_iq4nl / kvalues_mxfp4 (ggml-common.h:1088-1096)
    const uint32_t T0 = NL ? 0xbfad9881u : 0x03020100u, T1 = NL ? 0xf6eaddcfu : 0x0c080604u, T2 = NL ? 0x26190d01u : 0xfdfeff00u, T3 = NL ? 0x71594535u : 0xf4f8fafcu;
    for (int b = 0; b < a.nblk; b++) {
        const char * blk = wr + (int64_t) b * BS;
        const uint32_t q4 = MX ? ld1x4(blk + QOFF + 4 * (A & 3)) : ld2(blk + QOFF + 4 * (A & 3));
        const uint32_t nib = (q4 >> ((A >> 2) * 4)) & 0x0f0f0f0fu;
        uint32_t w; float dw, mw = 0.0f;
        if (Q50 || Q51) {
            const uint32_t qh = ld2(blk + (Q50 ? 2 : 4));
            const uint32_t bits = (qh >> (4 * A)) & 0xfu;                      // the fifth bits of elements 4A..4A+3 -> bit 4 of bytes 0..3
            w = nib | (((bits * 0x00204081u) & 0x01010101u) << 4);
            dw = h2f(*(const uint16_t *) blk);
            if (Q51) mw = h2f(*(const uint16_t *)(blk + 2));
        } else {
            w = lut16(nib, T0, T1, T2, T3);
            dw = MX ? __uint_as_float(*(const uint8_t *) blk < 2 ? 0x00200000u << *(const uint8_t *) blk : (uint32_t)(*(const uint8_t *) blk - 1) << 23)      // ggml_e8m0_to_fp32_half (ggml-impl.h:471-489)
                    : h2f(*(const uint16_t *) blk);
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int cc = c < a.ncols ? c : 0;
            const char * ar = actb + (int64_t) cc * a.act_stride;
            const uint32_t av = *(const uint32_t *)(ar + b * 32 + 4 * A);
            const int sumi = Q50 ? dot4(w, av, dot4(0xf0f0f0f0u, av, 0)) : dot4(w, av, 0);      // Q5_0: (q5 - 16) . a
            const float yd = ((const float *)(ar + a.off_d))[b];
            const float d = (NL || MX) ? yd * dw : dw * yd;
            if (b < npair) {
                if (CHAIN || !(b & 1)) acc[c] = __builtin_fmaf(d, (float) sumi, acc[c]); else acc2[c] = __builtin_fmaf(d, (float) sumi, acc2[c]);
                if (Q51) summs[c] = __builtin_fmaf(mw, ((const float *)(ar + a.off_s))[b], summs[c]);      // summs += m_w * s_a, one fma in the reference build
            } else summs[c] = d * (float) group8_sum_i(sumi);               // the unpaired last block: sumf += d * (sumi1 + sumi2) after the horizontal sum
        }
    }
#pragma unroll
    for (int c = 0; c < NC; c++) {
        float v = CHAIN ? acc[c] : acc[c] + acc2[c];
        v = v + __int_as_float(lane_xor4_i(__float_as_int(v)));
        v = v + dpp_f<DPP_QUAD_XOR2>(v);
        v = v + dpp_f<DPP_QUAD_XOR1>(v);
        if (Q51 || (!CHAIN && (a.nblk & 1))) v = v + summs[c];
        if (live && A == 0 && c < a.ncols) dstb[(int64_t) c * a.ldd + row] = v;
    }
}

// act: the act rows (launch_quantize_act, kind act_kind_of(wtype)) of the M columns; dst[m * ldd + n]
static int kq_check(int wtype, const tview & w, int64_t K, int64_t N) {
    const bool k256 = is_k256_type(wtype);
    if (!is_kq_type(wtype) || K % (k256 ? 256 : 32) || N <= 0 || N > (1 << 28)) return CLLM_E_UNSUPPORTED;
    const uintptr_t al = (uintptr_t) w.data | (uintptr_t) w.nb[1] | (uintptr_t) w.nb[2];
    if (wtype == CLLM_TYPE_Q5_K ? (al & 15) : wtype == CLLM_TYPE_IQ4_XS ? (al & 3) : wtype == CLLM_TYPE_MXFP4 ? 0 : (al & 1)) return CLLM_E_UNSUPPORTED;
    return CLLM_OK;
}
static void kq_launch(hipStream_t st, int wtype, const kq_args & a, dim3 grid, bool chain) {
#define GO(T) do { if (a.ncols == 1) hipLaunchKernelGGL((k_gemv_kq<T, 1>), grid, dim3(256), 0, st, a); \
                   else if (a.ncols <= 4) hipLaunchKernelGGL((k_gemv_kq<T, 4>), grid, dim3(256), 0, st, a); \
                   else hipLaunchKernelGGL((k_gemv_kq<T, 8>), grid, dim3(256), 0, st, a); } while (0)
#define GB(T, CH) do { if (a.ncols == 1) hipLaunchKernelGGL((k_gemv_b32<T, 1, CH>), grid, dim3(256), 0, st, a); \
                       else if (a.ncols <= 4) hipLaunchKernelGGL((k_gemv_b32<T, 4, CH>), grid, dim3(256), 0, st, a); \
                       else hipLaunchKernelGGL((k_gemv_b32<T, 8, CH>), grid, dim3(256), 0, st, a); } while (0)
    switch (wtype) {
        case CLLM_TYPE_Q5_K: GO(CLLM_TYPE_Q5_K); break;
        case CLLM_TYPE_Q6_K: GO(CLLM_TYPE_Q6_K); break;
        case CLLM_TYPE_Q2_K: GO(CLLM_TYPE_Q2_K); break;
        case CLLM_TYPE_Q3_K: GO(CLLM_TYPE_Q3_K); break;
        case CLLM_TYPE_IQ4_XS: GO(CLLM_TYPE_IQ4_XS); break;
        case CLLM_TYPE_TQ1_0: GO(CLLM_TYPE_TQ1_0); break;
        case CLLM_TYPE_TQ2_0: GO(CLLM_TYPE_TQ2_0); break;
        case CLLM_TYPE_IQ2_XXS: GO(CLLM_TYPE_IQ2_XXS); break;
        case CLLM_TYPE_IQ2_XS: GO(CLLM_TYPE_IQ2_XS); break;
        case CLLM_TYPE_IQ2_S: GO(CLLM_TYPE_IQ2_S); break;
        case CLLM_TYPE_IQ3_XXS: GO(CLLM_TYPE_IQ3_XXS); break;
        case CLLM_TYPE_IQ3_S: GO(CLLM_TYPE_IQ3_S); break;
        case CLLM_TYPE_IQ1_S: GO(CLLM_TYPE_IQ1_S); break;
        case CLLM_TYPE_IQ1_M: GO(CLLM_TYPE_IQ1_M); break;
        case CLLM_TYPE_Q5_0: GB(CLLM_TYPE_Q5_0, true); break;
        case CLLM_TYPE_Q5_1: GB(CLLM_TYPE_Q5_1, true); break;
        case CLLM_TYPE_IQ4_NL: if (chain) GB(CLLM_TYPE_IQ4_NL, true); else GB(CLLM_TYPE_IQ4_NL, false); break;
        default: GB(CLLM_TYPE_MXFP4, false); break;
    }
#undef GO
#undef GB
}
int launch_gemv_kq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, int64_t M, float * dst, int64_t ldd) {
    const int64_t K = w.ne[0], N = w.ne[1];
    if (kq_check(wtype, w, K, N)) return CLLM_E_UNSUPPORTED;
    const int kind = act_kind_of(wtype);
    kq_args a = {};
    a.W = w.data; a.nb01 = w.nb[1]; a.N = (int) N; a.nblk = (int)(K / (is_k256_type(wtype) ? 256 : 32));
    a.act_stride = (int64_t) act_stride; a.off_d = (int) act_off_d(K); a.off_s = (int) act_off_s(K, kind);
    a.ldd = ldd;
    const dim3 grid((unsigned)((N * 8 + 255) / 256));
    const bool chain = M >= 2;                           // IQ4_NL: llamafile_sgemm serves n >= 2 (sgemm.cpp:3691, 4000-4013)
    for (int64_t m0 = 0; m0 < M; m0 += 8) {
        a.act = (const char *) act + (size_t) m0 * act_stride; a.dst = dst + m0 * ldd; a.ncols = (int)(M - m0 < 8 ? M - m0 : 8);
        kq_launch(st, wtype, a, grid, chain);
        LAUNCH_CHECK();
    }
    return CLLM_OK;
}
// MUL_MAT_ID with coverage-type expert weights: as [K, N, n_as], act rows of b (row index = i11 + ne11 * token), ids [n_used, n_tok] I32 in device memory, dst [N, n_used, n_tok].
// One grid slice per (token, slot): always the one-column order (mul_mat_id calls vec_dot: IQ4_NL's paired accumulators)
int launch_gemv_kq_id(hipStream_t st, int wtype, const tview & as, const void * act, size_t act_stride, int64_t ne11, const tview & ids, const tview & dst) {
    const int64_t K = as.ne[0], N = as.ne[1], n_used = ids.ne[0], n_tok = ids.ne[1];
    if (kq_check(wtype, as, K, N) || n_used * n_tok > 65535 || n_used <= 0 || ids.nb[0] != 4 || ids.nb[1] % 4 || dst.nb[1] % 4 || dst.nb[2] % 4) return CLLM_E_UNSUPPORTED;
    const int kind = act_kind_of(wtype);
    kq_args a = {};
    a.W = as.data; a.nb01 = as.nb[1]; a.N = (int) N; a.nblk = (int)(K / (is_k256_type(wtype) ? 256 : 32));
    a.act = (const char *) act; a.act_stride = (int64_t) act_stride; a.off_d = (int) act_off_d(K); a.off_s = (int) act_off_s(K, kind);
    a.dst = (float *) dst.data; a.ldd = (int64_t)(dst.nb[1] / 4); a.ncols = 1;
    a.ids = (const int32_t *) ids.data; a.ids_nb1 = (int64_t)(ids.nb[1] / 4); a.nb02 = as.nb[2]; a.dst_tok = (int64_t)(dst.nb[2] / 4); a.n_used = (int) n_used; a.ne11 = (int) ne11;
    kq_launch(st, wtype, a, dim3((unsigned)((N * 8 + 255) / 256), (unsigned)(n_used * n_tok)), false);
    LAUNCH_CHECK();
    return CLLM_OK;
}
