// dense_f16.hip -- OPT-IN prefill contraction on the fp16 matrix cores: dequantize -> dense GEMM (north star: "MFMA tiles ... for the dense
// bf16/fp16 prefill contraction"; SURVEY 7.1 step 5 path B; the reference's own CUDA backend does the same above its mmq limits,
// ggml-cuda.cu:1229, 2262-2265: dequantize the weights, convert the activations, call the BLAS GEMM).
//
//   CLLM_PREFILL=f16   W (Q4_0 / Q4_1 / Q8_0 / Q4_K) -> fp16 [N, K] (dequantize_row_*'s values, rounded to fp16), X f32 -> fp16 (RNE),
//                      D[N, M] = W . X^T in fp32 by the library GEMM (rocBLAS gemm_ex, f16 inputs, f32 accumulate: a plain dense GEMM is
//                      what the vendor library is for)
//
// This is NOT the reference CPU computation (which quantizes the activations to Q8_0 / Q8_K and takes integer block dot products): no
// activation-quantization error, an fp16 rounding of the weights instead -- a different, usually slightly MORE accurate result.  It is
// therefore off by default (the default prefill is the exact-integer int8-MFMA path, mmq.hip, parity tier T1; short prompts take the
// bit-exact mat-vec path); DESIGN.md states the measured deviation next to the measured TFLOP/s.
#include "common.h"
#include "dequant.h"

#include <dlfcn.h>
#include <rocblas/rocblas.h>        // types and enums only: the library is dlopen'ed on first use (it is large, and a process that also hosts
                                    // PyTorch -- bench.py at N > 1 -- already carries another copy)
static rocblas_handle g_rb = nullptr;
static rocblas_status (*p_create)(rocblas_handle *) = nullptr;
static rocblas_status (*p_set_stream)(rocblas_handle, hipStream_t) = nullptr;
static rocblas_status (*p_gemm_ex)(rocblas_handle, rocblas_operation, rocblas_operation, rocblas_int, rocblas_int, rocblas_int, const void *, const void *, rocblas_datatype,
                                   rocblas_int, const void *, rocblas_datatype, rocblas_int, const void *, const void *, rocblas_datatype, rocblas_int, void *, rocblas_datatype,
                                   rocblas_int, rocblas_datatype, rocblas_gemm_algo, int32_t, uint32_t) = nullptr;
static int load_rocblas() {
    if (p_gemm_ex) return CLLM_OK;
    void * h = dlopen("librocblas.so.5", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librocblas.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) FAIL(CLLM_E_UNSUPPORTED, "dense_f16: librocblas not found (%s)", dlerror());
    p_create = (decltype(p_create)) dlsym(h, "rocblas_create_handle");
    p_set_stream = (decltype(p_set_stream)) dlsym(h, "rocblas_set_stream");
    p_gemm_ex = (decltype(p_gemm_ex)) dlsym(h, "rocblas_gemm_ex");
    if (!p_create || !p_set_stream || !p_gemm_ex) { p_gemm_ex = nullptr; FAIL(CLLM_E_UNSUPPORTED, "dense_f16: rocBLAS symbols missing"); }
    return CLLM_OK;
}
static void * g_w16 = nullptr, * g_x16 = nullptr;
static size_t g_w16_bytes = 0, g_x16_bytes = 0;

__global__ void __launch_bounds__(256) k_dequant_f16(int type, const char * __restrict__ w, int64_t nb1, int64_t K, int64_t nrows, uint16_t * __restrict__ out) {
    const int64_t e = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 2;
    const int64_t row = blockIdx.y;
    if (e >= K) return;
    const char * r = w + row * nb1;
    const uint32_t lo = f2h(dequant_elem(type, r, e)), hi = f2h(dequant_elem(type, r, e + 1));
    *(uint32_t *)(out + row * K + e) = lo | (hi << 16);
}
__global__ void __launch_bounds__(256) k_f32_to_f16(const char * __restrict__ x, int64_t nb1, int64_t K, uint16_t * __restrict__ out) {
    const int64_t e = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 2;
    const int64_t row = blockIdx.y;
    if (e >= K) return;
    const float * r = (const float *)(x + row * nb1);
    *(uint32_t *)(out + row * K + e) = (uint32_t) f2h(r[e]) | ((uint32_t) f2h(r[e + 1]) << 16);
}

static int ensure(void *& p, size_t & have, size_t need, hipStream_t st) {
    if (need <= have) return CLLM_OK;
    HIP_TRY(hipStreamSynchronize(st));
    if (p) (void) hipFree(p);
    p = nullptr; have = 0;
    HIP_TRY(hipMalloc(&p, need));
    have = need;
    return CLLM_OK;
}

bool prefill_f16_enabled() { static const bool v = getenv("CLLM_PREFILL") && !strcmp(getenv("CLLM_PREFILL"), "f16"); return v; }

// w: [K, N] quantized rows (2-D), x: [K, M] f32 rows (nb1 stride), d: [N, M] f32 (nb1 stride); CLLM_E_UNSUPPORTED -> the int8 path takes it
int launch_dense_f16(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d) {
    const int64_t K = w.ne[0], N = w.ne[1], M = x.ne[1];
    if (K % 2 || d.nb[1] % 4 || N > INT32_MAX || M > INT32_MAX || K > INT32_MAX) return CLLM_E_UNSUPPORTED;
    if (int rc = load_rocblas()) return rc;
    if (!g_rb) { if (p_create(&g_rb) != rocblas_status_success) FAIL(CLLM_E_HIP, "dense_f16: rocblas_create_handle failed"); }
    if (p_set_stream(g_rb, st) != rocblas_status_success) FAIL(CLLM_E_HIP, "dense_f16: rocblas_set_stream failed");
    if (int rc = ensure(g_w16, g_w16_bytes, (size_t) N * K * 2, st)) return rc;
    if (int rc = ensure(g_x16, g_x16_bytes, (size_t) M * K * 2, st)) return rc;
    hipLaunchKernelGGL(k_dequant_f16, dim3((unsigned)((K / 2 + 255) / 256), (unsigned) N), dim3(256), 0, st, wtype, (const char *) w.data, w.nb[1], K, N, (uint16_t *) g_w16);
    hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)((K / 2 + 255) / 256), (unsigned) M), dim3(256), 0, st, (const char *) x.data, x.nb[1], K, (uint16_t *) g_x16);
    LAUNCH_CHECK();
    // column-major view: W16 is (K x N) with ld K, X16 is (K x M) with ld K, D is (N x M) with ld nb1 / 4:  D = W16^T . X16
    const float alpha = 1.0f, beta = 0.0f;
    const rocblas_status rs = p_gemm_ex(g_rb, rocblas_operation_transpose, rocblas_operation_none, (rocblas_int) N, (rocblas_int) M, (rocblas_int) K, &alpha,
                                              g_w16, rocblas_datatype_f16_r, (rocblas_int) K, g_x16, rocblas_datatype_f16_r, (rocblas_int) K, &beta,
                                              d.data, rocblas_datatype_f32_r, (rocblas_int)(d.nb[1] / 4), d.data, rocblas_datatype_f32_r, (rocblas_int)(d.nb[1] / 4),
                                              rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    if (rs != rocblas_status_success) FAIL(CLLM_E_HIP, "dense_f16: rocblas_gemm_ex failed (%d)", (int) rs);
    return CLLM_OK;
}
