// gemv_decode.hip -- the launchers of the decode step's quantized mat-vec (the kernel: gemv_decode_kernel.h).  The sparse-MoE forms (router, down + combine)
// are instantiated in gemv_moe.hip: their code stays out of this code object, whose layout the decode step's launch-to-launch time is sensitive to
// (the same kernels, bit for bit, measured 1.8 % slower per decode step with the MoE instantiations placed among them).
#include "gemv_decode_kernel.h"

static unsigned long long * g_gemv_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_mmvq_ts(unsigned long long * dev_buf) { g_gemv_ts = dev_buf; }   // tools only

// K a multiple of the block size, K <= 16384 (32768 for the plain-quantize prologue), nrows * row bytes < 4 GiB; returns
// CLLM_E_UNSUPPORTED for shapes the general kernels must take
int launch_gemv_decode(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps,
                       int epi, float * dst, const float * bias, const float * resid, const float * padd, float * xout) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype)) return CLLM_E_UNSUPPORTED;
    if ((wtype != CLLM_TYPE_Q4_K || decode_free_order() >= 2) && !padd && !g_gemv_ts && decode_free_order()) {      // opt-in: the free-order tier of the 32-weight block formats (gemv_free32.hip)
        const int rc = launch_gemv_decode_free(st, wtype, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (wtype == CLLM_TYPE_Q4_K && !padd && !g_gemv_ts) {          // the LDS-staged form (gemv_rows.hip) where the shape suits it
        const int rc = launch_gemv_rows(st, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (wtype != CLLM_TYPE_Q4_K && !padd && !g_gemv_ts) {          // Q4_0 / Q4_1 / Q8_0: the LDS-staged one-lane-group-per-row form (gemv_rows32.hip)
        int rc = launch_gemv_rows32(st, wtype, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
        rc = launch_gemv_team32(st, wtype, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid);      // few rows per CU: teams of waves per 8 rows (gemv_team32.hip)
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (K % kind || K > ((pro == 2 || pro == 4) ? 32768 : 16384) || pro < 1 || pro > 4 || nrows <= 0 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024) return CLLM_E_UNSUPPORTED;
    if (padd && (pro != 1 || K > 4096 || !xout || xout == px)) return CLLM_E_UNSUPPORTED;
    if (epi == 1 && (pro != 1 || nrows % 2 || (nrows / 2) % 8 || bias || resid)) FAIL(CLLM_E_UNSUPPORTED, "gemv_decode: SiLU epilogue needs gate/up row pairs, features %% 8 == 0");
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    int64_t grid = (units + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / kind);
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);     // + the waves' chain records
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GO3(FMT_, PRO_, EPI_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, PRO_, EPI_, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, PRO_, EPI_, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, padd, (const char *) W, nblk, kfull, nrem, eps, dst, xout, bias, resid, g_gemv_ts, \
                           (const int32_t *) nullptr, 0ull, 0, 0); } while (0)
#define GO(FMT_) do { \
        if (pro == 1 && epi == 1) { if (npre == 1) GO3(FMT_, 1, 1, 1); else GO3(FMT_, 1, 1, 4); } \
        else if (pro == 1)        { if (npre == 1) GO3(FMT_, 1, 0, 1); else GO3(FMT_, 1, 0, 4); } \
        else if (pro == 2)        { if (npre == 1) GO3(FMT_, 2, 0, 1); else if (npre == 4) GO3(FMT_, 2, 0, 4); else GO3(FMT_, 2, 0, 8); } \
        else if (pro == 4)        { if (npre == 1) GO3(FMT_, 4, 0, 1); else if (npre == 4) GO3(FMT_, 4, 0, 4); else GO3(FMT_, 4, 0, 8); } \
        else                      { if (npre == 1) GO3(FMT_, 3, 0, 1); else GO3(FMT_, 3, 0, 4); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GO(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GO(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GO(CLLM_TYPE_Q4_1); else GO(CLLM_TYPE_Q8_0);
#undef GO
#undef GO3
    LAUNCH_CHECK();
    return CLLM_OK;
}

// MUL_MAT_ID for ONE token: n_slots x (dst[:, slot] = W_expert(ids[slot]) . quantize(px + slot * px_slot_stride)), the activation quantized inside
// the kernel (prologue 2) -- one launch instead of quantize + mat-vec, and the decode kernel's streaming.  CLLM_E_UNSUPPORTED: general path.
// epi 1: every expert's rows alternate gate_u, up_u (cllm_pack_rows, interleave); dst[u, slot] = silu(gate_u . x) * (up_u . x), u < nrows / 2
int launch_gemv_decode_id(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, int64_t px_slot_stride,
                          const int32_t * ids, int n_slots, float * dst, int64_t dst_slot_stride, int epi) {
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype) || K % kind || K > 32768 || nrows <= 0 || n_slots < 1 || n_slots > 64 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024 || px_slot_stride > INT32_MAX || dst_slot_stride > INT32_MAX) return CLLM_E_UNSUPPORTED;
    if (epi != 0 && (epi != 1 || nrows % 2 || (nrows / 2) % 8)) return CLLM_E_UNSUPPORTED;
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    int64_t grid = (units + 15) / 16;
    int64_t cap = device_cu_count() / n_slots; if (cap < 1) cap = 1;
    if (grid > cap) grid = cap;
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / kind);
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOM(FMT_, EPI_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, 2, EPI_, NPRE_, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, 2, EPI_, NPRE_, true>), dim3((unsigned) grid, (unsigned) n_slots), dim3(1024), lds, st, px, (const float *) nullptr, (const float *) nullptr, (const char *) W, \
                           nblk, kfull, nrem, 0.0f, dst, (float *) nullptr, (const float *) nullptr, (const float *) nullptr, (unsigned long long *) nullptr, ids, \
                           (unsigned long long) w_expert_bytes, (int) px_slot_stride, (int) dst_slot_stride); } while (0)
#define GOMT(FMT_) do { if (epi == 1) { if (npre == 1) GOM(FMT_, 1, 1); else if (npre == 4) GOM(FMT_, 1, 4); else GOM(FMT_, 1, 8); } \
                        else          { if (npre == 1) GOM(FMT_, 0, 1); else if (npre == 4) GOM(FMT_, 0, 4); else GOM(FMT_, 0, 8); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOMT(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOMT(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOMT(CLLM_TYPE_Q4_1); else GOMT(CLLM_TYPE_Q8_0);
#undef GOMT
#undef GOM
    LAUNCH_CHECK();
    return CLLM_OK;
}

