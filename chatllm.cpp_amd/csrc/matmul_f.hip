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

// ---- accumulation ORDER: the reference's, lane for lane, so that the results are bit-identical to libggml-cpu.so ----------------------
//   VD (ggml_vec_dot_f16 / _f32, vec.cpp:264- / 11-, AVX2: simd-mappings.h:528-620): 32 fp32 accumulators, accumulator a takes elements
//      a, a + 32, a + 64, ... in order with one fma each; GGML_F32x8_REDUCE's tree; the K mod 32 leftovers one by one (F16: in double,
//      F32: a float fma).  Taken for one src1 column, or when tinyBLAS declines.
//   T8 (tinyBLAS<8, __m256, ...>, llamafile/sgemm.cpp:477-640): with >= 2 src1 columns, K % 8 == 0 and ne01 % 4 == 0 the reference runs
//      ONE 8-lane accumulator per output element over steps of 8, then hsum.
// Mapping: VD -- 16 lanes per src0 row, lane c carries accumulators 2c and 2c + 1; T8 -- 8 lanes per row, lane l carries accumulator l.
// Four src1 columns share every src0 load.  (The many-column prefill contractions run on the matrix cores, mma_f16.hip: tolerance tier.)
#include "q4k.h"        // lane_xor4_i, DPP_ROW_ROR8
__device__ __forceinline__ float lane_xor4_f(float v) { return __int_as_float(lane_xor4_i(__float_as_int(v))); }

struct mmf_cols { const char * xc[4]; float * dc[4]; bool ok[4]; };
__device__ __forceinline__ void mmf_columns(const tview & x, const tview & d, int64_t c0, int64_t ncol, int64_t i02, int64_t i03, int64_t r2, int64_t r3, mmf_cols & C) {
#pragma unroll
    for (int c = 0; c < 4; c++) {
        int64_t q = c0 + c; C.ok[c] = q < ncol; if (!C.ok[c]) q = c0;
        const int64_t i11 = q % x.ne[1]; q /= x.ne[1];
        const int64_t i12 = i02 * r2 + q % r2, i13 = i03 * r3 + q / r2;
        C.xc[c] = x.data + i11*x.nb[1] + i12*x.nb[2] + i13*x.nb[3];
        C.dc[c] = (float *)(d.data + i11*d.nb[1] + i12*d.nb[2] + i13*d.nb[3]);
    }
}

// grid.x over groups of 16 src0 rows (i01), grid.y = ne02*ne03
template <typename T>
__global__ void __launch_bounds__(256) k_mul_mat_f_vd(tview w, tview x, tview d) {
    const int tid = threadIdx.x, c16 = tid & 15;
    const int64_t i01 = (int64_t) blockIdx.x * 16 + (tid >> 4);
    const int64_t i02 = blockIdx.y % w.ne[2], i03 = blockIdx.y / w.ne[2];
    const bool active = i01 < w.ne[1];
    const int64_t K = w.ne[0], np = K & ~(int64_t) 31;
    const int64_t r2 = x.ne[2] / w.ne[2], r3 = x.ne[3] / w.ne[3];
    const char * wrow = w.data + (active ? i01 : 0) * w.nb[1] + i02 * w.nb[2] + i03 * w.nb[3];
    const int64_t ncol = x.ne[1] * r2 * r3;                   // src1 columns that use this src0 slice
    for (int64_t c0 = 0; c0 < ncol; c0 += 4) {
        mmf_cols C;
        mmf_columns(x, d, c0, ncol, i02, i03, r2, r3, C);
        float a0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) for (int64_t i = 0; i < np; i += 32) {
            const int64_t e = i + 2 * c16;
            const float w0 = ld8<T>::one(wrow + e * sizeof(T)), w1 = ld8<T>::one(wrow + (e + 1) * sizeof(T));
#pragma unroll
            for (int c = 0; c < 4; c++) {
                a0[c] = __builtin_fmaf(w0, ld8<T>::cvt(*(const float *)(C.xc[c] + e * 4)), a0[c]);
                a1[c] = __builtin_fmaf(w1, ld8<T>::cvt(*(const float *)(C.xc[c] + e * 4 + 4)), a1[c]);
            }
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            // accumulator a = 8j + l sits in lane (a / 2): x0 += x2, x1 += x3 (lanes c ^ 8); x0 += x1 (c ^ 4); lo + hi (c ^ 2); hadd, hadd
            float v0 = a0[c], v1 = a1[c];
            v0 = v0 + dpp_f<DPP_ROW_ROR8>(v0); v1 = v1 + dpp_f<DPP_ROW_ROR8>(v1);
            v0 = v0 + lane_xor4_f(v0);         v1 = v1 + lane_xor4_f(v1);
            v0 = v0 + dpp_f<DPP_QUAD_XOR2>(v0); v1 = v1 + dpp_f<DPP_QUAD_XOR2>(v1);
            float u = v0 + v1;
            u = u + dpp_f<DPP_QUAD_XOR1>(u);
            if (active && c16 == 0 && C.ok[c]) {
                if (sizeof(T) == 2) {
                    double s = (double) u;
                    for (int64_t e = np; e < K; e++) s += (double)(ld8<T>::one(wrow + e * sizeof(T)) * ld8<T>::cvt(*(const float *)(C.xc[c] + e * 4)));
                    u = (float) s;
                } else {
                    // `sumf += x[i]*y[i]` as gcc -O3 compiles it (determined against libggml-cpu.so): whole groups of 4 leftovers have their
                    // products rounded (vector multiply) and added in order; the scalar remainder is contracted into fmas
                    int64_t e = np;
                    for (; e + 4 <= K; e += 4) for (int l = 0; l < 4; l++) u = u + ld8<T>::one(wrow + (e + l) * sizeof(T)) * *(const float *)(C.xc[c] + (e + l) * 4);
                    for (; e < K; e++) u = __builtin_fmaf(ld8<T>::one(wrow + e * sizeof(T)), *(const float *)(C.xc[c] + e * 4), u);
                }
                C.dc[c][i01] = u;
            }
        }
    }
}

// grid.x over groups of 32 src0 rows, grid.y = ne02*ne03; K % 8 == 0
template <typename T>
__global__ void __launch_bounds__(256) k_mul_mat_f_t8(tview w, tview x, tview d) {
    const int tid = threadIdx.x, l = tid & 7;
    const int64_t i01 = (int64_t) blockIdx.x * 32 + (tid >> 3);
    const int64_t i02 = blockIdx.y % w.ne[2], i03 = blockIdx.y / w.ne[2];
    const bool active = i01 < w.ne[1];
    const int64_t K = w.ne[0];
    const int64_t r2 = x.ne[2] / w.ne[2], r3 = x.ne[3] / w.ne[3];
    const char * wrow = w.data + (active ? i01 : 0) * w.nb[1] + i02 * w.nb[2] + i03 * w.nb[3];
    const int64_t ncol = x.ne[1] * r2 * r3;
    for (int64_t c0 = 0; c0 < ncol; c0 += 4) {
        mmf_cols C;
        mmf_columns(x, d, c0, ncol, i02, i03, r2, r3, C);
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (active) for (int64_t k = l; k < K; k += 8) {
            const float wv = ld8<T>::one(wrow + k * sizeof(T));
#pragma unroll
            for (int c = 0; c < 4; c++) acc[c] = __builtin_fmaf(wv, ld8<T>::cvt(*(const float *)(C.xc[c] + k * 4)), acc[c]);
        }
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float v = acc[c];                                  // hsum: x[i] + x[4 + i]; [0] + [2], [1] + [3]; [0] + [1]
            v = v + lane_xor4_f(v);
            v = v + dpp_f<DPP_QUAD_XOR2>(v);
            v = v + dpp_f<DPP_QUAD_XOR1>(v);
            if (active && l == 0 && C.ok[c]) C.dc[c][i01] = v;
        }
    }
}

template <typename T>
static int launch_T(hipStream_t st, const tview & w, const tview & x, const tview & d) {
    const int64_t K = w.ne[0];
    if (w.ne[2] * w.ne[3] > 65535) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_f: too many batches");
    // llamafile_sgemm takes the product when n >= 2, k % 8 == 0, m % 4 == 0 (sgemm.cpp:3691, 488, 503-517); else the vec_dot loop
    const bool t8 = x.ne[1] >= 2 && K % 8 == 0 && w.ne[1] % 4 == 0;
    if (t8) hipLaunchKernelGGL((k_mul_mat_f_t8<T>), dim3((unsigned)((w.ne[1] + 31) / 32), (unsigned)(w.ne[2] * w.ne[3])), dim3(256), 0, st, w, x, d);
    else    hipLaunchKernelGGL((k_mul_mat_f_vd<T>), dim3((unsigned)((w.ne[1] + 15) / 16), (unsigned)(w.ne[2] * w.ne[3])), dim3(256), 0, st, w, x, d);
    LAUNCH_CHECK();
    return CLLM_OK;
}

int launch_mul_mat_f(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d, int causal, int n_past) {
    // many src1 columns (prefill attention): matrix cores (mma_f16.hip); few columns (decode): the lane-group mat-vec below
    static const int mma_min_cols = getenv("CLLM_MMA_MIN_COLS") ? atoi(getenv("CLLM_MMA_MIN_COLS")) : 33;      // (below: the exact-order kernels, so that prompts of <= 32 tokens stay bit-identical to the CPU path)
    // (the exact-order matrix-core kernel has the lane-group kernels' bits: it takes over where it is faster -- CLLM_MMF_EXACT_MIN_COLS)
    static const int mmf_min_cols = getenv("CLLM_MMF_EXACT_MIN_COLS") ? atoi(getenv("CLLM_MMF_EXACT_MIN_COLS")) : 2;         // measured: faster from 2 columns on (profiles/r03_short_prompt_crossover.txt)
    const bool fast = wtype == CLLM_TYPE_F16 && x.ne[1] >= mma_min_cols && prefill_attn_mode() != 1;
    if (fast || (wtype == CLLM_TYPE_F16 && x.ne[1] >= mmf_min_cols)) {
        const int rc = fast ? launch_mma_f16(st, w, x, d, causal, n_past)
                            : launch_mmf_exact(st, w, x, d, causal, n_past);       // the reference's order on the f32 matrix cores (mmf_exact.hip)
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (wtype == CLLM_TYPE_F16) return launch_T<uint16_t>(st, w, x, d);
    if (wtype == CLLM_TYPE_F32) return launch_T<float>(st, w, x, d);
    FAIL(CLLM_E_UNSUPPORTED, "mul_mat_f: type %d", wtype);
}
