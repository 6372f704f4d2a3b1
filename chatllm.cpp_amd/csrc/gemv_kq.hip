// gemv_kq.hip -- MUL_MAT with Q5_K / Q6_K weights (the other k-quants third-party GGMM files carry: Q4_K_M / Q5_K_M mixes keep some
// tensors in Q5_K and Q6_K), for any number of columns, BIT-IDENTICAL to the reference's x86 AVX2 dot products:
//   ggml_vec_dot_q5_K_q8_K (ggml-cpu/arch/x86/quants.c:1916-2030)   8 lane accumulators acc[A] = fma(y.d x.d, (float) sumi[A], acc[A]) per
//                                                                    super-block in order, + ONE scalar chain summs += (-y.d x.dmin) * sum m S
//   ggml_vec_dot_q6_K_q8_K (ggml-cpu/arch/x86/quants.c:2130-2225)   the same lanes, int8 scale per 16 elements, no mins
//   result = hsum_float_8(acc) [+ summs]
// AVX lane A = dword A of every 32-byte chunk.  The simple, obviously-exact mapping: 8 GPU lanes per weight row, lane A computes sumi[A] of every
// super-block itself and carries acc[A] in a register, so the serial fp32 chain needs no cross-lane traffic at all; the three adds of
// hsum_float_8 are lane exchanges at the end.  Activations: the Q8_K act rows of quantize.hip (common.h layout).  Q6_K blocks are 210 bytes
// (2-byte aligned): two 16-bit loads per dword.  This is the coverage path (it streams at a fraction of the Q4_K kernels' rate), not a tuned one.
#include "common.h"
#include "q4k.h"


struct kq_args {
    const char * W; int64_t nb01; int N, nblk;
    const char * act; int64_t act_stride; int off_d, off_s;
    float * dst; int64_t ldd; int ncols;
};

__device__ __forceinline__ uint32_t ld2(const char * p) { return (uint32_t) *(const uint16_t *) p | ((uint32_t) *(const uint16_t *)(p + 2) << 16); }

template <int TYPE, int NC>
__global__ void __launch_bounds__(256) k_gemv_kq(const kq_args a) {
    const int tid = blockIdx.x * 256 + threadIdx.x, A = tid & 7;
    int row = tid >> 3;
    const bool live = row < a.N;                       // whole waves stay: the final lane exchanges read their neighbours
    if (!live) row = a.N - 1;
    const char * wr = a.W + (int64_t) row * a.nb01;
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
            const char * ar = a.act + (int64_t) cc * a.act_stride;
            int sumi = 0;
#pragma unroll
            for (int t = 0; t < 8; t++) sumi += (int) sc8[t] * dot4(w[t], *(const uint32_t *)(ar + b * 256 + aoff[t]), 0);
            const float yd = ((const float *)(ar + a.off_d))[b];
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
        if (live && A == 0 && c < a.ncols) a.dst[(int64_t) c * a.ldd + row] = v;
    }
}

// act: Q8_K act rows (launch_quantize_act, kind ACT_Q8_K) of the M columns; dst[m * ldd + n]
int launch_gemv_kq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, int64_t M, float * dst, int64_t ldd) {
    const int64_t K = w.ne[0], N = w.ne[1];
    if ((wtype != CLLM_TYPE_Q5_K && wtype != CLLM_TYPE_Q6_K) || K % 256 || N <= 0 || N > (1 << 28)) return CLLM_E_UNSUPPORTED;
    if (wtype == CLLM_TYPE_Q5_K ? (((uintptr_t) w.data | (uintptr_t) w.nb[1]) & 15) : (((uintptr_t) w.data | (uintptr_t) w.nb[1]) & 1)) return CLLM_E_UNSUPPORTED;
    kq_args a;
    a.W = w.data; a.nb01 = w.nb[1]; a.N = (int) N; a.nblk = (int)(K / 256);
    a.act_stride = (int64_t) act_stride; a.off_d = (int) act_off_d(K); a.off_s = (int) act_off_s(K, ACT_Q8_K);
    a.ldd = ldd;
    const dim3 grid((unsigned)((N * 8 + 255) / 256));
    for (int64_t m0 = 0; m0 < M; m0 += 8) {
        a.act = (const char *) act + (size_t) m0 * act_stride; a.dst = dst + m0 * ldd; a.ncols = (int)(M - m0 < 8 ? M - m0 : 8);
#define GO(T) do { if (a.ncols == 1) hipLaunchKernelGGL((k_gemv_kq<T, 1>), grid, dim3(256), 0, st, a); \
                   else if (a.ncols <= 4) hipLaunchKernelGGL((k_gemv_kq<T, 4>), grid, dim3(256), 0, st, a); \
                   else hipLaunchKernelGGL((k_gemv_kq<T, 8>), grid, dim3(256), 0, st, a); } while (0)
        if (wtype == CLLM_TYPE_Q5_K) GO(CLLM_TYPE_Q5_K); else GO(CLLM_TYPE_Q6_K);
#undef GO
        LAUNCH_CHECK();
    }
    return CLLM_OK;
}
