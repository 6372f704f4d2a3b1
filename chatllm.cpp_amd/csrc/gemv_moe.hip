// gemv_moe.hip -- the sparse-MoE forms of the decode mat-vec (k_gemv_dec EPI 2: router, EPI 3: down projection + combine), instantiated apart from
// the dense decode kernels (see gemv_decode.hip).
#include "gemv_decode_kernel.h"

// The router of a sparse-MoE block for ONE token as one launch of one workgroup (the reference's nodes RMS_NORM -> MUL -> MUL_MAT(gate) -> SOFT_MAX ->
// TOP_K, GenericSparseMLP::forward src/layers.cpp:3792-3830): xnorm[K] = RMS_NORM(px) * pw, probs[n] = SOFT_MAX(W . quantize(xnorm)), ids[k] = TOP_K(probs).
// Same reductions in the same order as the separate kernels: bit-identical.  n <= 64 experts, K <= 16384; CLLM_E_UNSUPPORTED otherwise.
int launch_moe_router(hipStream_t st, int wtype, const void * W, int64_t K, int64_t n, const float * px, const float * pw, float eps,
                      float * xnorm, float * probs, int32_t * ids, int k) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || K % kind || K % 4 || K > 16384 || n < 1 || n > 64 || k < 1 || k > n) return CLLM_E_UNSUPPORTED;
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024) return CLLM_E_UNSUPPORTED;
    const int kfull = (int)(n / 16), nrem = (int)(n % 16), nblk = (int)(K / kind);
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES) + 2 * 64 * sizeof(float);     // + logits, probabilities
#define GOR(FMT_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, 1, 2, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, 1, 2, NPRE_>), dim3(1), dim3(1024), lds, st, px, pw, (const float *) nullptr, (const char *) W, nblk, kfull, nrem, eps, probs, xnorm, \
                           (const float *) nullptr, (const float *) nullptr, (unsigned long long *) nullptr, (const int32_t *) ids, 0ull, 0, k); } while (0)
#define GORT(FMT_) do { if (K <= 4096) GOR(FMT_, 1); else GOR(FMT_, 4); } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GORT(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GORT(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GORT(CLLM_TYPE_Q4_1); else GORT(CLLM_TYPE_Q8_0);
#undef GORT
#undef GOR
    LAUNCH_CHECK();
    return CLLM_OK;
}

// MUL_MAT_ID(down experts) for ONE token with TWO slots + the tail of the sparse-MoE block in one launch (EPI 3 above):
//   dst[r] = (W[ids[0]][r] . quantize(px[:, 0])) * w0 + (W[ids[1]][r] . quantize(px[:, 1])) * w1 (+ resid[r]),  w_j = probs[ids[j]] / (probs[ids[0]] + probs[ids[1]])
// the arithmetic of MUL_MAT_ID -> GET_ROWS -> SUM_ROWS -> DIV -> MUL -> ADD (-> ADD) in their order: bit-identical.  dst may be resid; CLLM_E_UNSUPPORTED otherwise
int launch_gemv_decode_id_combine(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, int64_t px_slot_stride,
                                  const int32_t * ids, const float * probs, const float * resid, float * dst) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || K % kind || K > 32768 || nrows <= 0 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32) || px_slot_stride > INT32_MAX || px_slot_stride % 4) return CLLM_E_UNSUPPORTED;
    const size_t lds = 2 * act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);
    if (2 * act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024) return CLLM_E_UNSUPPORTED;
    int64_t grid = (nrows + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(nrows / nwaves), nrem = (int)(nrows % nwaves), nblk = (int)(K / kind);
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOC(FMT_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, 2, 3, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, 2, 3, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, probs, (const float *) nullptr, (const char *) W, nblk, kfull, nrem, 0.0f, dst, (float *) nullptr, \
                           (const float *) nullptr, resid, (unsigned long long *) nullptr, ids, (unsigned long long) w_expert_bytes, (int) px_slot_stride, 0); } while (0)
#define GOCT(FMT_) do { if (npre == 1) GOC(FMT_, 1); else if (npre == 4) GOC(FMT_, 4); else GOC(FMT_, 8); } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOCT(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOCT(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOCT(CLLM_TYPE_Q4_1); else GOCT(CLLM_TYPE_Q8_0);
#undef GOCT
#undef GOC
    LAUNCH_CHECK();
    return CLLM_OK;
}

// The head of a sparse-MoE block AND its experts' gate / up projections for ONE token as one launch (EPI 5 above): RMS_NORM -> MUL -> MUL_MAT(router) -> SOFT_MAX -> TOP_K ->
// {MUL_MAT_ID(gate), MUL_MAT_ID(up)} -> SiLU -> MUL (GenericSparseMLP::forward src/layers.cpp:3792-3872) -- every workgroup redoes the router behind its norm prologue, slot
// blockIdx.y streams the rows of expert ids[slot].  probs[ne] / ids[k] are published for the down + combine launch.  W: the per-expert interleaved gate / up pack (rows 2u = gate_u,
// 2u + 1 = up_u), nrows = 2 F; dst[u + slot * dst_slot_stride].  The same reductions in the same order as the separate launches: bit-identical.  CLLM_E_UNSUPPORTED otherwise.
int launch_gemv_decode_id_router_silu(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, const float * pw, float eps,
                                      const void * Wr, int ne, int k, float * probs, int32_t * ids, float * dst, int64_t dst_slot_stride) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || K % kind || K % 4 || K > 16384 || ne < 1 || ne > 64 || k < 1 || k > ne || nrows <= 0 || nrows % 2 || (nrows / 2) % 8 ||
        (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32) || dst_slot_stride > INT32_MAX) return CLLM_E_UNSUPPORTED;
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES) + 3 * 64 * sizeof(float);     // + logits, probabilities, ids
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES + 3 * 64 * sizeof(float) > 160 * 1024) return CLLM_E_UNSUPPORTED;
    const int64_t units = nrows / 2;
    int64_t grid = (units + 15) / 16;
    int64_t cap = device_cu_count() / k; if (cap < 1) cap = 1;
    if (grid > cap) grid = cap;
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / kind);
#define GOX(FMT_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, 1, 5, NPRE_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, 1, 5, NPRE_, true>), dim3((unsigned) grid, (unsigned) k), dim3(1024), lds, st, px, pw, (const float *) Wr, (const char *) W, nblk, kfull, nrem, eps, dst, probs, \
                           (const float *) nullptr, (const float *) nullptr, (unsigned long long *) nullptr, (const int32_t *) ids, (unsigned long long) w_expert_bytes, ne, (int) dst_slot_stride); } while (0)
#define GOXT(FMT_) do { if (K <= 4096) GOX(FMT_, 1); else GOX(FMT_, 4); } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOXT(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOXT(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOXT(CLLM_TYPE_Q4_1); else GOXT(CLLM_TYPE_Q8_0);
#undef GOXT
#undef GOX
    LAUNCH_CHECK();
    return CLLM_OK;
}
