// mma_f16.hip -- MUL_MAT with an F16 src0 and many F32 src1 columns on the matrix cores: the prefill attention
// contractions K.Q (scores) and V.P (context) of the eager (non-flash) attention path (src/layers.cpp:2540-2633).
//
//   dst[n, m] = sum_k f16(src0[k, n]) * f16r(src1[k, m])        ggml_vec_dot_f16 semantics (ggml-cpu.c:1190-1245 picks
//   vec_dot_type F16 for an F16 src0: src1 is rounded to fp16, products are exact in fp32, fp32 accumulation)
//
// Both operands are contiguous along k, which is exactly the fragment shape of v_mfma_f32_16x16x32_f16 (a lane holds 8
// consecutive k of one row / column), so fragments are loaded straight from global memory -- no LDS transpose.  src1 plays
// the MFMA "A" role (i = m) and src0 the "B" role (j = n): the D fragment then has 16 consecutive n per 16 lanes, i.e.
// 64-byte contiguous stores into dst columns.  A 256-thread workgroup owns a 128 (n) x 128 (m) tile, each wave 64 x 64.
// These two GEMMs are bound by the fp32 score matrix traffic ([n_kv, qlen, heads]: 2 GB per layer at 4096 tokens), not by
// MFMA throughput; `causal` lets the runner skip what the mask removes anyway:
//   causal 1 (K.Q): a tile whose every (kv position n, query m) has n > n_past + m is not computed (soft_max never reads it);
//   causal 2 (V.P): the k loop of a query tile stops at n_past + m_hi + 1 (P is exactly 0 beyond).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

struct mma_args {
    const char * a0; int64_t nb01, nb02, nb03;       // src0 (F16): row n, batch strides (bytes)
    const char * b0; int64_t nb11, nb12, nb13;       // src1 (F32): column m, batch strides
    char * d0;       int64_t nb1, nb2, nb3;          // dst  (F32)
    int N, M, K;                                     // src0 rows, src1 columns, shared dimension
    int ne12, r2, r3;                                // batch: blockIdx.z = i12 + ne12 * i13; src0 batch = (i12 / r2, i13 / r3)
    int causal, n_past;
};

__global__ void __launch_bounds__(256) k_mma_f16(const mma_args a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 128 + (wave & 1) * 64, m0 = blockIdx.y * 128 + (wave >> 1) * 64;
    const int i12 = blockIdx.z % a.ne12, i13 = blockIdx.z / a.ne12;
    int kend = a.K;
    if (a.causal == 1 && (int) blockIdx.x * 128 > a.n_past + (int) blockIdx.y * 128 + 127) return;
    if (a.causal == 2) kend = min(a.K, a.n_past + (int) blockIdx.y * 128 + 128);
    if (n0 >= a.N || m0 >= a.M) return;
    const char * A = a.a0 + (int64_t)(i12 / a.r2) * a.nb02 + (int64_t)(i13 / a.r3) * a.nb03;
    const char * B = a.b0 + (int64_t) i12 * a.nb12 + (int64_t) i13 * a.nb13;
    char *       D = a.d0 + (int64_t) i12 * a.nb2 + (int64_t) i13 * a.nb3;

    const int l15 = lane & 15, kq = (lane >> 4) * 8;
    f32x4v acc[4][4];                                   // [tile along m][tile along n]
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = f32x4v{0, 0, 0, 0};

    const char * arow[4]; const char * bcol[4]; bool aok[4], bok[4];
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int n = n0 + t * 16 + l15, m = m0 + t * 16 + l15;
        aok[t] = n < a.N; bok[t] = m < a.M;
        arow[t] = A + (int64_t)(aok[t] ? n : 0) * a.nb01;
        bcol[t] = B + (int64_t)(bok[t] ? m : 0) * a.nb11;
    }
    for (int k0 = 0; k0 < kend; k0 += 32) {
        const int k = k0 + kq;
        const bool kfull = k + 8 <= kend;
        half8 fa[4], fb[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {                   // src0 rows: 8 halves = one 16-byte load
            u32x4 r = u32x4{0, 0, 0, 0};
            if (aok[t] && kfull) r = *(const u32x4 *)(arow[t] + (int64_t) k * 2);
            else if (aok[t] && k < kend) {              // ragged tail of k
                uint16_t h[8];
#pragma unroll
                for (int e = 0; e < 8; e++) h[e] = k + e < kend ? *(const uint16_t *)(arow[t] + (int64_t)(k + e) * 2) : (uint16_t) 0;
                r = u32x4{ (uint32_t) h[0] | ((uint32_t) h[1] << 16), (uint32_t) h[2] | ((uint32_t) h[3] << 16), (uint32_t) h[4] | ((uint32_t) h[5] << 16), (uint32_t) h[6] | ((uint32_t) h[7] << 16) };
            }
            fb[t] = __builtin_bit_cast(half8, r);
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {                   // src1 columns: 8 floats, rounded to fp16 (RNE, as GGML_FP32_TO_FP16)
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = 0.0f;
            if (bok[t] && kfull) {
                const f32x4 lo = *(const f32x4 *)(bcol[t] + (int64_t) k * 4), hi = *(const f32x4 *)(bcol[t] + (int64_t) k * 4 + 16);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else if (bok[t] && k < kend) {
#pragma unroll
                for (int e = 0; e < 8; e++) if (k + e < kend) v[e] = *(const float *)(bcol[t] + (int64_t)(k + e) * 4);
            }
            half8 f;
#pragma unroll
            for (int e = 0; e < 8; e++) f[e] = (_Float16) v[e];
            fa[t] = f;
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    // D fragment: column (n) = lane & 15, row (m) = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = n0 + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int m = m0 + i * 16 + (lane >> 4) * 4 + r;
                if (n < a.N && m < a.M) *(float *)(D + (int64_t) m * a.nb1 + (int64_t) n * 4) = acc[i][j][r];
            }
        }
}

// CLLM_E_UNSUPPORTED: shape/alignment this kernel does not take (the caller falls back to k_mul_mat_f)
int launch_mma_f16(hipStream_t st, const tview & w, const tview & x, const tview & d, int causal, int n_past) {
    if (w.nb[0] != 2 || x.nb[0] != 4 || d.nb[0] != 4) return CLLM_E_UNSUPPORTED;
    if (((uintptr_t) w.data | (uintptr_t) w.nb[1] | (uintptr_t) w.nb[2] | (uintptr_t) w.nb[3]) & 15) return CLLM_E_UNSUPPORTED;
    if (((uintptr_t) x.data | (uintptr_t) x.nb[1] | (uintptr_t) x.nb[2] | (uintptr_t) x.nb[3]) & 15) return CLLM_E_UNSUPPORTED;
    if (w.ne[0] > INT32_MAX || w.ne[1] > INT32_MAX || x.ne[1] > INT32_MAX || x.ne[2] * x.ne[3] > 65535) return CLLM_E_UNSUPPORTED;
    mma_args a;
    a.a0 = w.data; a.nb01 = w.nb[1]; a.nb02 = w.nb[2]; a.nb03 = w.nb[3];
    a.b0 = x.data; a.nb11 = x.nb[1]; a.nb12 = x.nb[2]; a.nb13 = x.nb[3];
    a.d0 = d.data; a.nb1 = d.nb[1]; a.nb2 = d.nb[2]; a.nb3 = d.nb[3];
    a.N = (int) w.ne[1]; a.M = (int) x.ne[1]; a.K = (int) w.ne[0];
    a.ne12 = (int) x.ne[2]; a.r2 = (int)(x.ne[2] / w.ne[2]); a.r3 = (int)(x.ne[3] / w.ne[3]);
    a.causal = causal; a.n_past = n_past;
    const dim3 grid((unsigned)((a.N + 127) / 128), (unsigned)((a.M + 127) / 128), (unsigned)(x.ne[2] * x.ne[3]));
    hipLaunchKernelGGL(k_mma_f16, grid, dim3(256), 0, st, a);
    LAUNCH_CHECK();
    return CLLM_OK;
}
