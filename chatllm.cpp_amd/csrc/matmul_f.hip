// matmul_f.hip -- MUL_MAT with F16 / F32 src0 (the attention contractions K.Q and V.P and any float weights).
//
// Reference: ggml_compute_forward_mul_mat with type F16/F32 (ggml-cpu/ggml-cpu.c:1229-1421): src1 rows are first
// converted to src0's vec_dot_type (F16 for F16 weights: ggml-cpu.c:214-219, i.e. Q and the soft-max
// probabilities are rounded to fp16 with RNE), products are exact in fp32, accumulation is fp32
// (ggml_vec_dot_f16, vec.cpp:264-).  We reproduce the rounding of src1 and accumulate in fp32.
//
// src0 is read through its strides (K/V are permuted views of the cache with GQA broadcast:
// src/layers.cpp:3138-3144, 3172-3179): rows must be dense (nb[0] == element size).
//
// Mapping: a group of G lanes (G = 64 or less for short rows such as head_dim 128) owns one src0 row and keeps
// its 8-element slices in registers while it walks every src1 column that broadcasts onto that row
// (the r2 query heads of a GQA group x the ne11 tokens), 4 columns at a time.
#include "common.h"

template <typename T> struct ld8;
template <> struct ld8<uint16_t> {
    static __device__ __forceinline__ void load(const char * p, float (&v)[8]) {
        const u32x4 r = *(const u32x4 *) p;
        const uint32_t w[4] = { r.x, r.y, r.z, r.w };
#pragma unroll
        for (int i = 0; i < 4; i++) { v[2*i] = h2f((uint16_t)(w[i] & 0xffff)); v[2*i + 1] = h2f((uint16_t)(w[i] >> 16)); }
    }
    static __device__ __forceinline__ float one(const char * p) { return h2f(*(const uint16_t *) p); }
    static __device__ __forceinline__ float cvt(float y) { return h2f(f2h(y)); }      // src1 -> fp16 (RNE) -> fp32
};
struct __attribute__((packed, aligned(4))) f32x4_a4 { float x, y, z, w; };    // 4-byte aligned 16-byte load (rows of P are n_kv floats)
template <> struct ld8<float> {
    static __device__ __forceinline__ void load(const char * p, float (&v)[8]) {
        const f32x4_a4 a = *(const f32x4_a4 *) p, b = *(const f32x4_a4 *)(p + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    }
    static __device__ __forceinline__ float one(const char * p) { return *(const float *) p; }
    static __device__ __forceinline__ float cvt(float y) { return y; }
};

// grid.x over groups of src0 rows (i01), grid.y = ne02*ne03
template <typename T, int G, bool VEC>
__global__ void __launch_bounds__(256) k_mul_mat_f(tview w, tview x, tview d) {
    constexpr int NCB = 4;
    const int tid = threadIdx.x;
    const int gl = tid % G;                                   // lane inside the group
    const int64_t i01 = (int64_t) blockIdx.x * (256 / G) + tid / G;
    const int64_t i02 = blockIdx.y % w.ne[2], i03 = blockIdx.y / w.ne[2];
    const bool active = i01 < w.ne[1];
    const int64_t K = w.ne[0];
    const int64_t r2 = x.ne[2] / w.ne[2], r3 = x.ne[3] / w.ne[3];
    const char * wrow = w.data + (active ? i01 : 0) * w.nb[1] + i02 * w.nb[2] + i03 * w.nb[3];
    const int64_t ncol = x.ne[1] * r2 * r3;                   // src1 columns that use this src0 slice

    for (int64_t c0 = 0; c0 < ncol; c0 += NCB) {
        const char * xc[NCB]; float * dc[NCB]; bool ok[NCB];
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            int64_t q = c0 + c; ok[c] = q < ncol; if (!ok[c]) q = c0;
            const int64_t i11 = q % x.ne[1]; q /= x.ne[1];
            const int64_t i12 = i02 * r2 + q % r2, i13 = i03 * r3 + q / r2;
            xc[c] = x.data + i11*x.nb[1] + i12*x.nb[2] + i13*x.nb[3];
            dc[c] = (float *)(d.data + i11*d.nb[1] + i12*d.nb[2] + i13*d.nb[3]);
        }
        float acc[NCB] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) {
            if (VEC) {
                const int64_t K8 = K & ~(int64_t) 7;
                for (int64_t k = K8 + gl; k < K; k += G) {                 // ragged tail (n_kv is arbitrary for V.P)
                    const float wv = ld8<T>::one(wrow + k * sizeof(T));
#pragma unroll
                    for (int c = 0; c < NCB; c++) acc[c] = __builtin_fmaf(wv, ld8<T>::cvt(*(const float *)(xc[c] + k * 4)), acc[c]);
                }
                for (int64_t k = (int64_t) gl * 8; k < K8; k += G * 8) {
                    float wv[8];
                    ld8<T>::load(wrow + k * sizeof(T), wv);
#pragma unroll
                    for (int c = 0; c < NCB; c++) {
                        float xv[8];
                        ld8<float>::load(xc[c] + k * 4, xv);
#pragma unroll
                        for (int i = 0; i < 8; i++) acc[c] = __builtin_fmaf(wv[i], ld8<T>::cvt(xv[i]), acc[c]);
                    }
                }
            } else {
                for (int64_t k = gl; k < K; k += G) {
                    const float wv = ld8<T>::one(wrow + k * sizeof(T));
#pragma unroll
                    for (int c = 0; c < NCB; c++) acc[c] = __builtin_fmaf(wv, ld8<T>::cvt(*(const float *)(xc[c] + k * 4)), acc[c]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < NCB; c++) {
            float v = acc[c];
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (active && gl == 0 && ok[c]) dc[c][i01] = v;
        }
    }
}

template <typename T>
static int launch_T(hipStream_t st, const tview & w, const tview & x, const tview & d) {
    const int64_t K = w.ne[0];
    // vector path needs 16-byte aligned src0 rows; src1 rows only need their natural 4-byte alignment
    const bool vec = K >= 8 && ((uintptr_t) w.data % 16 == 0) && w.nb[1] % 16 == 0 && w.nb[2] % 16 == 0 && w.nb[3] % 16 == 0 &&
                     ((uintptr_t) x.data % 4 == 0) && x.nb[1] % 4 == 0 && x.nb[2] % 4 == 0 && x.nb[3] % 4 == 0;
    int G = 64;
    if (vec) { while (G > 8 && (int64_t) G * 8 > K) G >>= 1; } else { while (G > 8 && G > K) G >>= 1; }
    const int64_t rows_per_wg = 256 / G;
    dim3 grid((unsigned)((w.ne[1] + rows_per_wg - 1) / rows_per_wg), (unsigned)(w.ne[2] * w.ne[3]));
    if (grid.y > 65535) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_f: too many batches");
#define GO(GG) do { if (vec) hipLaunchKernelGGL((k_mul_mat_f<T, GG, true>), grid, dim3(256), 0, st, w, x, d); \
                    else     hipLaunchKernelGGL((k_mul_mat_f<T, GG, false>), grid, dim3(256), 0, st, w, x, d); } while (0)
    switch (G) { case 64: GO(64); break; case 32: GO(32); break; case 16: GO(16); break; default: GO(8); break; }
#undef GO
    LAUNCH_CHECK();
    return CLLM_OK;
}

int launch_mul_mat_f(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d, int causal, int n_past) {
    // many src1 columns (prefill attention): matrix cores (mma_f16.hip); few columns (decode): the lane-group mat-vec below
    static const int mma_min_cols = getenv("CLLM_MMA_MIN_COLS") ? atoi(getenv("CLLM_MMA_MIN_COLS")) : 32;
    if (wtype == CLLM_TYPE_F16 && x.ne[1] >= mma_min_cols) {
        const int rc = launch_mma_f16(st, w, x, d, causal, n_past);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (wtype == CLLM_TYPE_F16) return launch_T<uint16_t>(st, w, x, d);
    if (wtype == CLLM_TYPE_F32) return launch_T<float>(st, w, x, d);
    FAIL(CLLM_E_UNSUPPORTED, "mul_mat_f: type %d", wtype);
}
