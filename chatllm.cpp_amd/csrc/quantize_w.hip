// quantize_w.hip -- WEIGHT quantizers on the device: the reference's from_float_ref for Q8_0 / Q4_0 / Q4_1 / Q5_0 / Q5_1 / Q4_K and F16 (ggml-quants.c:36-222,
// 622-702, 1280-1350), byte-identical to libggml-base.so.  That is what chatllm.cpp's loader runs on the host, one tensor at a time, when a model file's tensor type
// differs from the one the run asks for (src/chat.cpp:1246-1279 -> ggml::from_float, src/layers.cpp:358-373: dequantize to fp32, quantize again, upload) -- here the
// fp32 rows are quantized where they will live (INTEGRATION.md shows the three-line change in the loader).
//
// The reference quantizers are plain scalar C compiled without contraction, so every operation is one IEEE rounding in the written order; the kernels keep that order:
//   * 32-weight blocks: one thread per block (the first element of largest magnitude decides sign and scale; min / max for the offset formats);
//   * Q4_K: 8 lanes per 256-weight super-block, lane j runs make_qkx2_quants for sub-block j (a weighted least-squares fit of scale and minimum over 21 candidate
//     rounding grids) on its 32 values in registers; the 6-bit scale / minimum codes are exchanged inside the lane group, every lane re-rounds its own quants with the
//     decoded scales, neighbouring lanes merge their nibbles.
// Divisions are IEEE (__fdiv_rn), the square root goes through double (exactly rounded to float).
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

template <int TYPE>
__global__ void __launch_bounds__(256) k_quantize_b32(const float * __restrict__ x, char * __restrict__ y, int64_t nblocks) {
    const int64_t b = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    float v[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { const f32x4 t = *(const f32x4 *)(x + b * 32 + 4 * i); v[4 * i] = t.x; v[4 * i + 1] = t.y; v[4 * i + 2] = t.z; v[4 * i + 3] = t.w; }
    constexpr bool SIGNED = TYPE == CLLM_TYPE_Q4_0 || TYPE == CLLM_TYPE_Q5_0 || TYPE == CLLM_TYPE_Q8_0;
    constexpr int BS = TYPE == CLLM_TYPE_Q8_0 ? 34 : TYPE == CLLM_TYPE_Q4_0 ? 18 : TYPE == CLLM_TYPE_Q4_1 ? 20 : TYPE == CLLM_TYPE_Q5_0 ? 22 : 24;
    uint8_t out[36];
    float d, mn = 0.0f;
    if (SIGNED) {
        float amax = 0.0f, mx = 0.0f;
#pragma unroll
        for (int j = 0; j < 32; j++) { const float a = fabsf(v[j]); if (amax < a) { amax = a; mx = v[j]; } }
        d = TYPE == CLLM_TYPE_Q8_0 ? __fdiv_rn(amax, 127.0f) : TYPE == CLLM_TYPE_Q4_0 ? __fdiv_rn(mx, -8.0f) : __fdiv_rn(mx, -16.0f);
    } else {
        float a = 3.402823466e+38f, m = -3.402823466e+38f;
#pragma unroll
        for (int j = 0; j < 32; j++) { if (v[j] < a) a = v[j]; if (v[j] > m) m = v[j]; }
        mn = a;
        d = __fdiv_rn(m - a, TYPE == CLLM_TYPE_Q4_1 ? 15.0f : 31.0f);
    }
    const float id = d != 0.0f ? __fdiv_rn(1.0f, d) : 0.0f;
    const uint16_t dh = f2h(d);
    out[0] = (uint8_t)(dh & 0xff); out[1] = (uint8_t)(dh >> 8);
    int o = 2;
    if (!SIGNED) { const uint16_t mh = f2h(mn); out[2] = (uint8_t)(mh & 0xff); out[3] = (uint8_t)(mh >> 8); o = 4; }
    if (TYPE == CLLM_TYPE_Q8_0) {
#pragma unroll
        for (int j = 0; j < 32; j++) out[2 + j] = (uint8_t)(int8_t) roundf(v[j] * id);          // roundf: halves away from zero (ggml-quants.c:199-222)
    } else {
        constexpr bool Q5 = TYPE == CLLM_TYPE_Q5_0 || TYPE == CLLM_TYPE_Q5_1;
        constexpr int QMAX = Q5 ? 31 : 15;
        const float off = SIGNED ? (Q5 ? 16.5f : 8.5f) : 0.5f;
        uint32_t qh = 0;
        const int qo = Q5 ? o + 4 : o;
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const float x0 = SIGNED ? v[j] * id : (v[j] - mn) * id, x1 = SIGNED ? v[j + 16] * id : (v[j + 16] - mn) * id;
            int q0 = (int)(x0 + off), q1 = (int)(x1 + off);                                      // (int8_t) of a value in 0 .. 32: the truncation
            if (TYPE != CLLM_TYPE_Q5_1) { q0 = q0 < QMAX ? q0 : QMAX; q1 = q1 < QMAX ? q1 : QMAX; }   // (Q5_1 does not clamp: (uint8_t)(x + 0.5f) <= 31)
            out[qo + j] = (uint8_t)((q0 & 0x0F) | ((q1 & 0x0F) << 4));
            if (Q5) { qh |= (uint32_t)((q0 >> 4) & 1) << j; qh |= (uint32_t)((q1 >> 4) & 1) << (j + 16); }
        }
        if (Q5) { out[o] = (uint8_t) qh; out[o + 1] = (uint8_t)(qh >> 8); out[o + 2] = (uint8_t)(qh >> 16); out[o + 3] = (uint8_t)(qh >> 24); }
    }
    uint16_t * yp = (uint16_t *)(y + b * BS);                                                     // blocks are 2-byte aligned
#pragma unroll
    for (int i = 0; i < BS / 2; i++) yp[i] = (uint16_t)(out[2 * i] | (out[2 * i + 1] << 8));
}

__global__ void __launch_bounds__(256) k_quantize_f16(const float * __restrict__ x, uint16_t * __restrict__ y, int64_t n) {
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i < n) y[i] = f2h(x[i]);
}

// ---- Q4_K: 8 lanes per super-block ----
__device__ __forceinline__ float sqrt_rn(float v) { return (float) sqrt((double) v); }           // exactly rounded (53 >= 2 * 24 + 2)
__device__ __forceinline__ float grp8_f(float v, int src, int lane) { return __shfl(v, (lane & ~7) | src, 64); }

__global__ void __launch_bounds__(256) k_quantize_q4_K(const float * __restrict__ x, char * __restrict__ y, int64_t nsb) {
    const int64_t t = (int64_t) blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, j = lane & 7;
    int64_t sb = t >> 3;
    const bool live = sb < nsb;                          // whole lane groups stay (the exchanges below)
    if (!live) sb = nsb - 1;
    float v[32];
#pragma unroll
    for (int i = 0; i < 8; i++) { const f32x4 q = *(const f32x4 *)(x + sb * 256 + 32 * j + 4 * i); v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w; }

    // weights of the fit: av_x + |x| (quantize_row_q4_K_ref, ggml-quants.c:1294-1298)
    float sum_x2 = 0.0f;
#pragma unroll
    for (int l = 0; l < 32; l++) sum_x2 = sum_x2 + v[l] * v[l];
    const float av_x = sqrt_rn(__fdiv_rn(sum_x2, 32.0f));
    float w[32];
#pragma unroll
    for (int l = 0; l < 32; l++) w[l] = av_x + fabsf(v[l]);

    // make_qkx2_quants(32, 15, x, w, L, &min, Laux, -1, 0.1, 20, false) (ggml-quants.c:622-702)
    int L[32], Laux[32];
    float mn = v[0], mx = v[0], sum_w = w[0], sum_x = sum_w * v[0];
#pragma unroll
    for (int i = 1; i < 32; i++) {
        if (v[i] < mn) mn = v[i];
        if (v[i] > mx) mx = v[i];
        sum_w = sum_w + w[i];
        sum_x = sum_x + w[i] * v[i];
    }
    if (mn > 0.0f) mn = 0.0f;
    float scale = 0.0f;
    if (mx == mn) {
#pragma unroll
        for (int i = 0; i < 32; i++) L[i] = 0;
    } else {
        float iscale = __fdiv_rn(15.0f, mx - mn);
        scale = __fdiv_rn(1.0f, iscale);
        float best = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            int l = nearest_int_dev(iscale * (v[i] - mn));
            l = l < 0 ? 0 : l > 15 ? 15 : l;
            L[i] = l;
            float diff = scale * (float) l + mn - v[i];
            diff = diff * diff;
            best = best + w[i] * diff;
        }
        for (int is = 0; is <= 20; is++) {
            iscale = __fdiv_rn(-1.0f + 0.1f * (float) is + 15.0f, mx - mn);
            float sum_l = 0.0f, sum_l2 = 0.0f, sum_xl = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; i++) {
                int l = nearest_int_dev(iscale * (v[i] - mn));
                l = l < 0 ? 0 : l > 15 ? 15 : l;
                Laux[i] = l;
                const float wl = w[i] * (float) l;
                sum_l = sum_l + wl;
                sum_l2 = sum_l2 + wl * (float) l;
                sum_xl = sum_xl + wl * v[i];
            }
            const float D = sum_w * sum_l2 - sum_l * sum_l;
            if (D > 0.0f) {
                float this_scale = __fdiv_rn(sum_w * sum_xl - sum_x * sum_l, D), this_min = __fdiv_rn(sum_l2 * sum_x - sum_l * sum_xl, D);
                if (this_min > 0.0f) { this_min = 0.0f; this_scale = __fdiv_rn(sum_xl, sum_l2); }
                float cur = 0.0f;
#pragma unroll
                for (int i = 0; i < 32; i++) {
                    float diff = this_scale * (float) Laux[i] + this_min - v[i];
                    diff = diff * diff;
                    cur = cur + w[i] * diff;
                }
                if (cur < best) {
#pragma unroll
                    for (int i = 0; i < 32; i++) L[i] = Laux[i];
                    best = cur; scale = this_scale; mn = this_min;
                }
            }
        }
    }
    const float the_min = -mn;

    // the super-block's codes: max over the eight sub-blocks (from 0), 6-bit codes, fp16 d / dmin
    const float max_scale = group8_max(fmaxf(scale, 0.0f)), max_min = group8_max(fmaxf(the_min, 0.0f));
    const float inv_scale = max_scale > 0.0f ? __fdiv_rn(63.0f, max_scale) : 0.0f, inv_min = max_min > 0.0f ? __fdiv_rn(63.0f, max_min) : 0.0f;
    int ls = nearest_int_dev(inv_scale * scale) & 0xff, lm = nearest_int_dev(inv_min * the_min) & 0xff;        // (uint8_t) of the int, then MIN(63, .)
    ls = ls < 63 ? ls : 63; lm = lm < 63 ? lm : 63;
    const uint16_t dh = f2h(__fdiv_rn(max_scale, 63.0f)), dmh = f2h(__fdiv_rn(max_min, 63.0f));
    char * yb = y + sb * 144;
    {
        int lsj[8], lmj[8];
#pragma unroll
        for (int s = 0; s < 8; s++) { lsj[s] = __shfl(ls, (lane & ~7) | s, 64); lmj[s] = __shfl(lm, (lane & ~7) | s, 64); }
        if (live && j == 0) {
            uint8_t sc[12];
#pragma unroll
            for (int s = 0; s < 4; s++) { sc[s] = (uint8_t) lsj[s]; sc[s + 4] = (uint8_t) lmj[s]; }
#pragma unroll
            for (int s = 4; s < 8; s++) {
                sc[s + 4] = (uint8_t)((lsj[s] & 0xF) | ((lmj[s] & 0xF) << 4));
                sc[s - 4] |= (uint8_t)((lsj[s] >> 4) << 6);
                sc[s]     |= (uint8_t)((lmj[s] >> 4) << 6);
            }
            u32x4 h;
            h.x = (uint32_t) dh | ((uint32_t) dmh << 16);
            h.y = sc[0] | (sc[1] << 8) | (sc[2] << 16) | ((uint32_t) sc[3] << 24);
            h.z = sc[4] | (sc[5] << 8) | (sc[6] << 16) | ((uint32_t) sc[7] << 24);
            h.w = sc[8] | (sc[9] << 8) | (sc[10] << 16) | ((uint32_t) sc[11] << 24);
            *(u32x4 *) yb = h;
        }
    }
    // re-round with the decoded scales (a zero scale keeps the search's quants)
    const float d = h2f(dh) * (float) ls;
    if (d != 0.0f) {
        const float dm = h2f(dmh) * (float) lm;
#pragma unroll
        for (int i = 0; i < 32; i++) {
            int l = nearest_int_dev(__fdiv_rn(v[i] + dm, d));
            L[i] = l < 0 ? 0 : l > 15 ? 15 : l;
        }
    }
    // 64 weights share 32 bytes: sub-block 2c in the low nibbles, 2c + 1 in the high ones
    uint32_t p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = (uint32_t) L[4 * i] | ((uint32_t) L[4 * i + 1] << 8) | ((uint32_t) L[4 * i + 2] << 16) | ((uint32_t) L[4 * i + 3] << 24);
#pragma unroll
    for (int i = 0; i < 8; i++) { const uint32_t other = (uint32_t) dpp_i<DPP_QUAD_XOR1>((int) p[i]); p[i] = (j & 1) ? 0u : (p[i] | (other << 4)); }
    if (live && !(j & 1)) {
        *(u32x4 *)(yb + 16 + 32 * (j >> 1))      = u32x4{p[0], p[1], p[2], p[3]};
        *(u32x4 *)(yb + 16 + 32 * (j >> 1) + 16) = u32x4{p[4], p[5], p[6], p[7]};
    }
}

// quantize nrows rows of k fp32 values (contiguous) into rows of `type` blocks (contiguous); x 16-byte aligned, k a multiple of the block size
extern "C" __attribute__((visibility("default")))
int cllm_op_quantize_rows(void * stream, int type, const float * x, void * y, int64_t k, int64_t nrows) {
    if (!x || !y || k <= 0 || nrows <= 0) FAIL(CLLM_E_INVALID, "quantize_rows: arguments");
    const int bs = cllm_blck_size(type);
    if (!bs || k % bs || ((uintptr_t) x & 15) || ((uintptr_t) y & 1)) FAIL(CLLM_E_INVALID, "quantize_rows: k %lld is not a multiple of the block (%d) or operand alignment", (long long) k, bs);
    hipStream_t st = (hipStream_t) stream;
    const int64_t n = k * nrows, nb = n / bs;
    const unsigned g = (unsigned)((nb + 255) / 256);
    if (nb > (int64_t) 0x7fffffff * 32) FAIL(CLLM_E_UNSUPPORTED, "quantize_rows: too many blocks in one call");
    switch (type) {
        case CLLM_TYPE_F16:  hipLaunchKernelGGL(k_quantize_f16, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, (uint16_t *) y, n); break;
        case CLLM_TYPE_Q8_0: hipLaunchKernelGGL((k_quantize_b32<CLLM_TYPE_Q8_0>), dim3(g), dim3(256), 0, st, x, (char *) y, nb); break;
        case CLLM_TYPE_Q4_0: hipLaunchKernelGGL((k_quantize_b32<CLLM_TYPE_Q4_0>), dim3(g), dim3(256), 0, st, x, (char *) y, nb); break;
        case CLLM_TYPE_Q4_1: hipLaunchKernelGGL((k_quantize_b32<CLLM_TYPE_Q4_1>), dim3(g), dim3(256), 0, st, x, (char *) y, nb); break;
        case CLLM_TYPE_Q5_0: hipLaunchKernelGGL((k_quantize_b32<CLLM_TYPE_Q5_0>), dim3(g), dim3(256), 0, st, x, (char *) y, nb); break;
        case CLLM_TYPE_Q5_1: hipLaunchKernelGGL((k_quantize_b32<CLLM_TYPE_Q5_1>), dim3(g), dim3(256), 0, st, x, (char *) y, nb); break;
        case CLLM_TYPE_Q4_K:
            if ((uintptr_t) y & 15) FAIL(CLLM_E_INVALID, "quantize_rows: Q4_K rows must be 16-byte aligned");
            hipLaunchKernelGGL(k_quantize_q4_K, dim3((unsigned)((nb * 8 + 255) / 256)), dim3(256), 0, st, x, (char *) y, nb); break;
        default: FAIL(CLLM_E_UNSUPPORTED, "quantize_rows: type %d has no device quantizer (Q8_0, Q4_0, Q4_1, Q5_0, Q5_1, Q4_K, F16)", type);
    }
    LAUNCH_CHECK();
    return CLLM_OK;
}
