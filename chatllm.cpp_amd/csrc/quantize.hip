// quantize.hip -- on-device activation quantizers (F32 -> Q8_0 / Q8_K "act rows", layout in common.h).
//
// Reproduces exactly what the reference CPU path does to src1 before its integer dot products:
//   Q8_0: ggml/src/ggml-cpu/arch/x86/quants.c:290-345  d = amax/127 (stored as fp16), id = 127/amax,
//         q = round-half-even(x*id)
//   Q8_K: ggml/src/ggml-quants.c:2555-2592             iscale = -127/max (max = FIRST element of largest |x|, signed),
//         q = min(127, nearest_int(iscale*x)), d = 1/iscale, bsums
// All results are bit-exact w.r.t. the CPU (checked in tests/test_quantize.py).
#include "common.h"
#include "quant_dev.h"

// 8 lanes per 32-block, 4 elements per lane, 8 blocks per wave.  Q81: the Q8_1 flavour (s plane = fp16(d * sum) as f32).
// SILU: the row holds 2 K interleaved (gate_e, up_e) pairs; the values quantized are silu(gate_e) * up_e (UNARY(SILU) + MUL of BaseMLP::forward)
template <bool SILU>
__device__ __forceinline__ f32x4 quant_src4(const float * x, int64_t e0, int64_t K) {
    if (!SILU) return *(const f32x4 *)(x + e0);
    const f32x4 p0 = *(const f32x4 *)(x + 2 * e0), p1 = *(const f32x4 *)(x + 2 * e0 + 4);      // (g0, u0, g1, u1), (g2, u2, g3, u3)
    const int64_t nv = K & ~(int64_t) 7;
    f32x4 v;
    v.x = silu_any(p0.x, e0 + 0 < nv) * p0.y; v.y = silu_any(p0.z, e0 + 1 < nv) * p0.w;
    v.z = silu_any(p1.x, e0 + 2 < nv) * p1.y; v.w = silu_any(p1.z, e0 + 3 < nv) * p1.w;
    return v;
}
template <bool Q81, bool SILU>
__global__ void __launch_bounds__(256) k_quantize_q8_0(const char * __restrict__ src, int64_t K, int64_t ne1, int64_t ne2,
                                                       int64_t nb1, int64_t nb2, int64_t nb3,
                                                       char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.y;                        // flattened i11 + ne11*(i12 + ne12*i13)
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *)(src + i1*nb1 + i2*nb2 + i3*nb3);
    const int64_t e0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e0 >= K) return;                                   // K % 32 == 0 -> whole 8-lane groups drop out together
    quant4_store<32, Q81>(act + row * act_stride, K, e0, threadIdx.x & 63, quant_src4<SILU>(x, e0, K));
}

// one wave per 256-block, 4 elements per lane.
template <bool SILU>
__global__ void __launch_bounds__(256) k_quantize_q8_K(const char * __restrict__ src, int64_t K, int64_t ne1, int64_t ne2,
                                                       int64_t nb1, int64_t nb2, int64_t nb3,
                                                       char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.y;
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *)(src + i1*nb1 + i2*nb2 + i3*nb3);
    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (blk * 256 >= K) return;
    const int64_t e0 = blk * 256 + lane * 4;
    quant4_store<256>(act + row * act_stride, K, e0, lane, quant_src4<SILU>(x, e0, K));
}

template <bool SILU>
static int quantize_act_impl(hipStream_t st, int kind, const tview & s, void * act, size_t act_stride) {
    const int kind_blk = act_blk(kind);
    const int64_t K = SILU ? s.ne[0] / 2 : s.ne[0];
    const int64_t rows = s.ne[1] * s.ne[2] * s.ne[3];
    if (K % kind_blk || (SILU && s.ne[0] % 2)) FAIL(CLLM_E_INVALID, "quantize_act: K=%lld not a multiple of %d", (long long) K, kind_blk);
    if (rows <= 0 || K <= 0) return CLLM_OK;
    if (rows > 65535) FAIL(CLLM_E_UNSUPPORTED, "quantize_act: too many rows (%lld)", (long long) rows);
    if (kind_blk == 32) {
        dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned) rows);
        if (kind == ACT_Q8_1) hipLaunchKernelGGL((k_quantize_q8_0<true, SILU>),  grid, dim3(256), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], (char *) act, act_stride);
        else                  hipLaunchKernelGGL((k_quantize_q8_0<false, SILU>), grid, dim3(256), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], (char *) act, act_stride);
    } else {
        dim3 grid((unsigned)((K / 256 + 3) / 4), (unsigned) rows);
        hipLaunchKernelGGL((k_quantize_q8_K<SILU>), grid, dim3(256), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], (char *) act, act_stride);
    }
    LAUNCH_CHECK();
    return CLLM_OK;
}
int launch_quantize_act(hipStream_t st, int kind, const tview & s, void * act, size_t act_stride) { return quantize_act_impl<false>(st, kind, s, act, act_stride); }
int launch_quantize_act_silu(hipStream_t st, int kind, const tview & s, void * act, size_t act_stride) { return quantize_act_impl<true>(st, kind, s, act, act_stride); }

// UNARY(SILU)(gate) -> MUL(up) -> quantize with gate and up as SEPARATE [K, rows] tensors (the reference's own graph: BaseMLP::forward src/layers.cpp:2475-2483);
// the values quantized are silu(gate) * up as the two element-wise nodes produce them (polynomial body below K & ~7, libm tail)
template <int KIND, bool Q81>
__global__ void __launch_bounds__(256) k_quantize_silu2(const char * __restrict__ gsrc, const char * __restrict__ usrc, int64_t K, int64_t g_nb1, int64_t u_nb1,
                                                        char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.y;
    const float * g = (const float *)(gsrc + row * g_nb1), * u = (const float *)(usrc + row * u_nb1);
    const int lane = threadIdx.x & 63;
    const int64_t e0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e0 >= K) return;                                   // K % KIND == 0 and KIND | 256: whole lane groups drop out together
    const f32x4 a = *(const f32x4 *)(g + e0), b = *(const f32x4 *)(u + e0);
    const int64_t nv = K & ~(int64_t) 7;
    f32x4 v;
    v.x = silu_any(a.x, e0 + 0 < nv) * b.x; v.y = silu_any(a.y, e0 + 1 < nv) * b.y; v.z = silu_any(a.z, e0 + 2 < nv) * b.z; v.w = silu_any(a.w, e0 + 3 < nv) * b.w;
    quant4_store<KIND, Q81>(act + row * act_stride, K, e0, lane, v);
}
int launch_quantize_act_silu2(hipStream_t st, int kind, const tview & g, const tview & u, void * act, size_t act_stride) {
    const int kind_blk = act_blk(kind);
    const int64_t K = g.ne[0], rows = g.ne[1];
    if (K % kind_blk || g.ne[2] != 1 || g.ne[3] != 1 || u.ne[0] != K || u.ne[1] != rows || u.ne[2] != 1 || u.ne[3] != 1 || ((uintptr_t) g.data & 15) || ((uintptr_t) u.data & 15) ||
        g.nb[1] % 16 || u.nb[1] % 16 || rows > 65535) return CLLM_E_UNSUPPORTED;
    if (rows <= 0 || K <= 0) return CLLM_OK;
    const dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned) rows);
    if (kind_blk == 32) {
        if (kind == ACT_Q8_1) hipLaunchKernelGGL((k_quantize_silu2<32, true>),  grid, dim3(256), 0, st, g.data, u.data, K, g.nb[1], u.nb[1], (char *) act, act_stride);
        else                  hipLaunchKernelGGL((k_quantize_silu2<32, false>), grid, dim3(256), 0, st, g.data, u.data, K, g.nb[1], u.nb[1], (char *) act, act_stride);
    } else hipLaunchKernelGGL((k_quantize_silu2<256, false>), grid, dim3(256), 0, st, g.data, u.data, K, g.nb[1], u.nb[1], (char *) act, act_stride);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// RMS_NORM -> MUL(weight) -> quantize of whole rows in one pass (the input_layernorm / post_attention_layernorm in front of a prefill MUL_MAT,
// LMBlock1Forward src/layers.cpp:2730-2760): one 1024-thread workgroup per row; the sum of squares, the scale and the two multiplications are those of
// k_rms_norm<true> (ops.hip) on the same values, the quantizer is the one above: the act row is bit-identical to the two launches.  K <= 16384, K % 4 == 0.
template <int KIND, bool Q81>
__global__ void __launch_bounds__(1024) k_rms_norm_quantize(const char * __restrict__ src, int64_t K, int64_t ne1, int64_t ne2, int64_t nb1, int64_t nb2, int64_t nb3,
                                                           const float * __restrict__ w, float eps, char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *)(src + i1*nb1 + i2*nb2 + i3*nb3);
    __shared__ double part[16];
    const int tid = threadIdx.x, lane = tid & 63;
    f32x4 vv[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int64_t e = (int64_t) tid * 4 + u * 4096; vv[u] = e < K ? *(const f32x4 *)(x + e) : f32x4{0, 0, 0, 0}; }
    const double sum = rms_block_sumsq_1024(x, K, vv[0], part);
    const float scale = rms_scale(sum, K, eps, x, nullptr, part);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const int64_t e = (int64_t) tid * 4 + u * 4096;
        if (e < K) {
            const f32x4 g = *(const f32x4 *)(w + e);
            f32x4 v = vv[u];
            v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w;
            quant4_store<KIND, Q81>(act + row * act_stride, K, e, lane, v);
        }
    }
}
int launch_quantize_act_norm(hipStream_t st, int kind, const tview & s, const float * norm_w, float eps, void * act, size_t act_stride) {
    const int kind_blk = act_blk(kind);
    const int64_t K = s.ne[0], rows = s.ne[1] * s.ne[2] * s.ne[3];
    if (K % kind_blk || K % 4 || K > 16384 || ((uintptr_t) s.data & 15) || s.nb[1] % 16 || s.nb[2] % 16 || s.nb[3] % 16 || ((uintptr_t) norm_w & 15)) return CLLM_E_UNSUPPORTED;
    if (rows <= 0 || K <= 0) return CLLM_OK;
    const dim3 grid((unsigned) rows);
    if (kind_blk == 32) {
        if (kind == ACT_Q8_1) hipLaunchKernelGGL((k_rms_norm_quantize<32, true>),  grid, dim3(1024), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], norm_w, eps, (char *) act, act_stride);
        else                  hipLaunchKernelGGL((k_rms_norm_quantize<32, false>), grid, dim3(1024), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], norm_w, eps, (char *) act, act_stride);
    } else hipLaunchKernelGGL((k_rms_norm_quantize<256, false>), grid, dim3(1024), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], norm_w, eps, (char *) act, act_stride);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ---- KAT surface: act row -> reference block layout ------------------------------------------------
__global__ void k_act_to_q8_0_blocks(const char * __restrict__ act, int64_t K, block_q8_0 * __restrict__ y) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K / 32) return;
    const int8_t * qs = (const int8_t *) act;
    const float *  dd = (const float *)(act + act_off_d(K));
    y[b].d = f2h(dd[b]);
    for (int j = 0; j < 32; j++) y[b].qs[j] = qs[b*32 + j];
}
__global__ void k_act_to_q8_1_blocks(const char * __restrict__ act, int64_t K, block_q8_1 * __restrict__ y) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K / 32) return;
    const int8_t * qs = (const int8_t *) act;
    const float *  dd = (const float *)(act + act_off_d(K));
    const float *  ss = (const float *)(act + act_off_s(K, ACT_Q8_1));
    y[b].d = f2h(dd[b]); y[b].s = f2h(ss[b]);
    for (int j = 0; j < 32; j++) y[b].qs[j] = qs[b*32 + j];
}
__global__ void k_act_to_q8_K_blocks(const char * __restrict__ act, int64_t K, block_q8_K * __restrict__ y) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K / 256) return;
    const int8_t * qs = (const int8_t *) act;
    const float *  dd = (const float *)(act + act_off_d(K));
    y[b].d = dd[b];
    for (int j = 0; j < 16; j++) {
        int s = 0;
        for (int i = 0; i < 16; i++) { const int8_t q = qs[b*256 + j*16 + i]; y[b].qs[j*16 + i] = q; s += q; }
        y[b].bsums[j] = (int16_t) s;
    }
}

static int quantize_to_blocks(void * stream, int kind, const float * x, void * y, int64_t k) {
    const int kind_blk = act_blk(kind);
    hipStream_t st = (hipStream_t) stream;
    if (!x || !y || k <= 0 || k % kind_blk) FAIL(CLLM_E_INVALID, "quantize_row: bad arguments");
    void * act = nullptr;
    const size_t bytes = act_row_bytes(k, kind);
    HIP_TRY(hipMalloc(&act, bytes));
    tview s; s.data = (char *) x; s.ne[0] = k; s.ne[1] = s.ne[2] = s.ne[3] = 1; s.nb[0] = 4; s.nb[1] = s.nb[2] = s.nb[3] = k * 4;
    int rc = launch_quantize_act(st, kind, s, act, bytes);
    if (rc == CLLM_OK) {
        const int64_t nb = k / kind_blk;
        if (kind == ACT_Q8_1) hipLaunchKernelGGL(k_act_to_q8_1_blocks, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, (const char *) act, k, (block_q8_1 *) y);
        else if (kind_blk == 32) hipLaunchKernelGGL(k_act_to_q8_0_blocks, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, (const char *) act, k, (block_q8_0 *) y);
        else                hipLaunchKernelGGL(k_act_to_q8_K_blocks, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, (const char *) act, k, (block_q8_K *) y);
        rc = cllm_hip_check(hipGetLastError(), "act_to_blocks", __FILE__, __LINE__);
    }
    (void) hipStreamSynchronize(st);
    (void) hipFree(act);
    return rc;
}

extern "C" int cllm_quantize_row_q8_0(void * stream, const float * x, void * y, int64_t k) { return quantize_to_blocks(stream, 32, x, y, k); }
extern "C" int cllm_quantize_row_q8_1(void * stream, const float * x, void * y, int64_t k) { return quantize_to_blocks(stream, ACT_Q8_1, x, y, k); }
extern "C" int cllm_quantize_row_q8_K(void * stream, const float * x, void * y, int64_t k) { return quantize_to_blocks(stream, 256, x, y, k); }
