// gemv_tp.hip -- the decode mat-vec's tensor-parallel forms with the all-reduce FUSED into the neighbouring launches (kernel: gemv_decode_kernel.h, EPI 4 / PRO 5).
//
// A tensor-parallel decoder layer all-reduces two [hidden] vectors (after o_proj and after down_proj; SURVEY.md 8(e): the reference has layer split only, SplitMethod::Row is a
// TODO at src/backend.h:322-327).  As launches of their own (RCCL, or tp_oneshot.hip's one-shot kernel) these are 2 of 7 launches per layer, each a kernel boundary + a cross-GPU
// round trip -- on an 8B model whose whole layer is ~42 us that alone forbids scaling.  Here there is NO all-reduce launch:
//   * the o / down mat-vec (EPI 4) sends every partial row result as an 8-byte {value, step number} granule into its rank's slot of EVERY rank's receive buffer
//     (one system-scope write-through store per rank, issued by lanes 0..nranks-1 of the wave that owns the row);
//   * the next mat-vec (PRO 5: qkv of the next layer, gate/up, the lm_head) polls the granules of all ranks in its OWN buffer until they carry this step's number, adds them in
//     rank order to the residual stream (every rank computes the same bits; two ranks: the bits of any other sum), and goes on with RMS_NORM -> quantize -> rows.
// Data-tagged granules need no flag, no fence and no barrier (MI355X_MICROARCH.md rows handoff-1to1 / transport-variants).  One slot per (site = 2 layer + {o, down}, rank): a rank
// cannot reach a site of step t + 1 before every peer has consumed that site of step t (it needs the peers' granules of the LAST site of step t), so slots are never overwritten
// under a reader.  The step number is a device word advanced by a one-thread launch at the start of every step (cllm_tp_fused_advance).
// These instantiations live in their own code object: the single-GPU kernels' layout (and launch-to-launch time) is untouched.
#include "gemv_decode_kernel.h"

static unsigned long long * g_tp_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_tp_ts(unsigned long long * dev_buf) { g_tp_ts = dev_buf; }   // tools only: stamps of the gather launches (8 per workgroup)

// partial rows of an o / down projection -> granules in every rank's receive buffer.  pro 2: quantize(px); pro 3: quantize(silu(gate) * up) on interleaved pairs
int launch_gemv_decode_tp_scatter(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const void * ctx_dev, int site) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || !ctx_dev || (pro != 2 && pro != 3)) return CLLM_E_UNSUPPORTED;
    if (K % kind || K > (pro == 2 ? 32768 : 16384) || nrows <= 0 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024) return CLLM_E_UNSUPPORTED;
    int64_t grid = (nrows + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(nrows / nwaves), nrem = (int)(nrows % nwaves), nblk = (int)(K / kind);
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOS(FMT_, PRO_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, PRO_, 4, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, PRO_, 4, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, (const float *) nullptr, (const float *) nullptr, (const char *) W, nblk, kfull, nrem, 0.0f, \
                           (float *) nullptr, (float *) nullptr, (const float *) nullptr, (const float *) nullptr, (unsigned long long *) nullptr, (const int32_t *) ctx_dev, 0ull, 0, site); } while (0)
#define GOSF(FMT_) do { \
        if (pro == 2) { if (npre == 1) GOS(FMT_, 2, 1); else if (npre == 4) GOS(FMT_, 2, 4); else GOS(FMT_, 2, 8); } \
        else          { if (npre == 1) GOS(FMT_, 3, 1); else GOS(FMT_, 3, 4); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOSF(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOSF(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOSF(CLLM_TYPE_Q4_1); else GOSF(CLLM_TYPE_Q8_0);
#undef GOSF
#undef GOS
    LAUNCH_CHECK();
    return CLLM_OK;
}

// RMS_NORM(px + all-reduced partials of `site`) * pw -> quantize -> rows; workgroup 0 stores the new residual stream to xout (!= px).  epi 0 (+ bias) / 1 (SiLU(gate) * up)
int launch_gemv_decode_tp_gather(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, const float * px, const float * pw, float eps, int epi, float * dst,
                                 const float * bias, const void * ctx_dev, int site, float * xout) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || !ctx_dev || !xout || xout == px || (epi != 0 && epi != 1)) return CLLM_E_UNSUPPORTED;
    if (K % kind || K > 16384 || nrows <= 0 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (epi == 1 && (nrows % 2 || (nrows / 2) % 8 || bias)) return CLLM_E_UNSUPPORTED;
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES) + (size_t) K * 4;      // + the row copy of the serial RMS fallback
    if (lds > 160 * 1024 - 256) return CLLM_E_UNSUPPORTED;
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    int64_t grid = (units + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / kind);
    const int npre = K <= 4096 ? 1 : 4;
#define GOG(FMT_, EPI_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, 5, EPI_, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, 5, EPI_, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const float *) nullptr, (const char *) W, nblk, kfull, nrem, eps, \
                           dst, xout, bias, (const float *) nullptr, g_tp_ts, (const int32_t *) ctx_dev, 0ull, site, 0); } while (0)
#define GOGF(FMT_) do { \
        if (epi == 1) { if (npre == 1) GOG(FMT_, 1, 1); else GOG(FMT_, 1, 4); } \
        else          { if (npre == 1) GOG(FMT_, 0, 1); else GOG(FMT_, 0, 4); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOGF(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOGF(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOGF(CLLM_TYPE_Q4_1); else GOGF(CLLM_TYPE_Q8_0);
#undef GOGF
#undef GOG
    LAUNCH_CHECK();
    return CLLM_OK;
}

// the all-reduce of `site` alone: xout = px + sum over ranks (rank order) of the granules -- the gather prologue's arithmetic without a mat-vec behind it (the last residual
// stream of a step whose head is not a fusable mat-vec: chatllm's LMFinalSteps keeps the normalised hidden state as a graph OUTPUT, src/models.cpp:1754-1755)
__global__ void __launch_bounds__(256) k_tpf_residual(const float * __restrict__ px, const tp_fuse_dev * __restrict__ cx, int site, int n, float * __restrict__ xout) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= n) return;
    const f32x4 g = tpf_gather4(cx, cx->peer[cx->rank], cx->nranks, cx->max_n, *cx->step, site, e);
    f32x4 v = *(const f32x4 *)(px + e);
    v.x = v.x + g.x; v.y = v.y + g.y; v.z = v.z + g.z; v.w = v.w + g.w;
    *(f32x4 *)(xout + e) = v;
}

// ---- C ABI (include/chatllm_hip.h): the two forms as operators of their own -- what the ggml module's logical tensor-parallel device issues once per rank (host/ggml-hip.cpp) ----
extern "C" const void * cllm_tp_fused_dev(void * os);
extern "C" int cllm_tp_fused_sites(void * os);
extern "C" size_t cllm_tp_fused_max_n(void * os);
// partial rows of a K-sharded o / down projection -> granules of `site` in every rank's receive buffer.  pro 2: quantize(px); pro 3: quantize(silu(gate) * up) over interleaved pairs
extern "C" CLLM_API int cllm_op_mul_mat_vec_tp_scatter(void * stream, const cllm_tensor * src0, int pro, const float * px, void * tp_fused, int site) {
    if (!src0 || !px || !tp_fused || (pro != 2 && pro != 3) || site < 0 || site >= cllm_tp_fused_sites(tp_fused)) FAIL(CLLM_E_INVALID, "mul_mat_vec_tp_scatter: arguments");
    if (!is_quant_type(src0->type) || src0->ne[2] != 1 || src0->ne[3] != 1 || src0->nb[1] != cllm_row_size(src0->type, src0->ne[0])) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_tp_scatter: src0 must be a dense 2-D quantized matrix");
    if ((size_t) src0->ne[1] > cllm_tp_fused_max_n(tp_fused)) FAIL(CLLM_E_INVALID, "mul_mat_vec_tp_scatter: %lld rows, the receive buffers hold %zu", (long long) src0->ne[1], cllm_tp_fused_max_n(tp_fused));
    if (((uintptr_t) px | (uintptr_t) src0->data) & 15) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_tp_scatter: alignment");
    return launch_gemv_decode_tp_scatter((hipStream_t) stream, src0->type, src0->data, src0->ne[0], src0->ne[1], pro, px, cllm_tp_fused_dev(tp_fused), site);
}
// dst = W . quantize(RMS_NORM(px + all-reduced partials of `site`) * pw) (+ bias | epi 1: SiLU(gate) * up over alternating rows); xout (!= px) receives the new residual stream
extern "C" CLLM_API int cllm_op_mul_mat_vec_tp_gather(void * stream, const cllm_tensor * src0, const float * px, const float * pw, float eps, int epi, const float * bias, float * dst,
                                                      void * tp_fused, int site, float * xout) {
    if (!src0 || !px || !pw || !dst || !xout || !tp_fused || (epi != 0 && epi != 1) || site < 0 || site >= cllm_tp_fused_sites(tp_fused)) FAIL(CLLM_E_INVALID, "mul_mat_vec_tp_gather: arguments");
    if (!is_quant_type(src0->type) || src0->ne[2] != 1 || src0->ne[3] != 1 || src0->nb[1] != cllm_row_size(src0->type, src0->ne[0])) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_tp_gather: src0 must be a dense 2-D quantized matrix");
    if ((size_t) src0->ne[0] > cllm_tp_fused_max_n(tp_fused)) FAIL(CLLM_E_INVALID, "mul_mat_vec_tp_gather: rows of %lld values, the receive buffers hold %zu", (long long) src0->ne[0], cllm_tp_fused_max_n(tp_fused));
    if (((uintptr_t) px | (uintptr_t) pw | (uintptr_t) src0->data | (uintptr_t) xout) & 15) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_tp_gather: alignment");
    return launch_gemv_decode_tp_gather((hipStream_t) stream, src0->type, src0->data, src0->ne[0], src0->ne[1], px, pw, eps, epi, dst, bias, cllm_tp_fused_dev(tp_fused), site, xout);
}
// xout[0..n) = px + the all-reduced partials of `site` (n % 4 == 0; xout may be px): the residual stream after the last down projection of a step
extern "C" CLLM_API int cllm_op_tp_gather_residual(void * stream, const float * px, int64_t n, void * tp_fused, int site, float * xout) {
    if (!px || !xout || !tp_fused || n <= 0 || n % 4 || site < 0 || site >= cllm_tp_fused_sites(tp_fused) || (size_t) n > cllm_tp_fused_max_n(tp_fused)) FAIL(CLLM_E_INVALID, "tp_gather_residual: arguments");
    if (((uintptr_t) px | (uintptr_t) xout) & 15) FAIL(CLLM_E_UNSUPPORTED, "tp_gather_residual: alignment");
    hipLaunchKernelGGL(k_tpf_residual, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t) stream, px, (const tp_fuse_dev *) cllm_tp_fused_dev(tp_fused), site, (int) n, xout);
    LAUNCH_CHECK();
    return CLLM_OK;
}
