// gemv_free32.hip -- the OPT-IN free-order tier of the decode mat-vec for the 32-weight block formats (Q4_0 / Q4_1 / Q8_0): CLLM_DECODE_FREE_ORDER=1 / cllm_set_decode_free_order(1).
//
// Everything of k_gemv_dec (gemv_decode_kernel.h: the activation built in the kernel, one row per wave, 64 blocks per step, the same prologues and epilogues) except the fp32
// fold: the exact int32 block dot products are scaled and added per lane (q32_block_free, q32.h) and the 64 lanes are summed by the wave at the end of the row -- NOT the order of
// ggml_vec_dot_q4_0_q8_0 / q4_1_q8_1 / q8_0_q8_0 (arch/x86/quants.c:543-577, 701-760, 1012-1040: eight per-AVX-lane fp32 chains over the blocks), which the default kernels
// (gemv_rows32.hip, gemv_team32.hip, k_gemv_dec) reproduce bit for bit.  It exists to PRICE that order (the review's "decide their contract": DESIGN.md section 6, round 6) and as
// a tolerance tier for hosts that ask for it; tests/test_gpu_llama.py states what it keeps (integer sums exact; ids and logits against the exact order at real shapes).
#include "gemv_decode_kernel.h"

static int g_free_order = -1;
int decode_free_order() { if (g_free_order < 0) g_free_order = getenv("CLLM_DECODE_FREE_ORDER") ? atoi(getenv("CLLM_DECODE_FREE_ORDER")) : 0; return g_free_order; }
extern "C" CLLM_API int cllm_set_decode_free_order(int on) { g_free_order = on < 0 ? 0 : on > 2 ? 2 : on; return CLLM_OK; }
extern "C" CLLM_API int cllm_get_decode_free_order(void) { return decode_free_order(); }

// the forms the decode step uses: pro 1 (RMS_NORM, + SiLU * up epilogue), pro 2 (plain quantize); anything else: CLLM_E_UNSUPPORTED -> the exact kernels
int launch_gemv_decode_free(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps,
                            int epi, float * dst, const float * bias, const float * resid) {
    // Q4_K: only with CLLM_DECODE_FREE_ORDER=2 -- a PRICING experiment (what the exact order costs the headline kernel), never a product tier: see q4k_block_free4
    if (wtype == CLLM_TYPE_Q4_K) { if (decode_free_order() < 2) return CLLM_E_UNSUPPORTED; }
    else if (wtype != CLLM_TYPE_Q4_0 && wtype != CLLM_TYPE_Q4_1 && wtype != CLLM_TYPE_Q8_0) return CLLM_E_UNSUPPORTED;
    const int kind = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if ((pro != 1 && pro != 2) || (epi != 0 && epi != 1) || (epi == 1 && pro != 1)) return CLLM_E_UNSUPPORTED;
    if (K % kind || K > (pro == 2 ? 32768 : 16384) || nrows <= 0 || (uint64_t) nrows * (uint64_t) cllm_row_size(wtype, K) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (act_row_bytes(K, kind) + 16 * Q32_CHAIN_BYTES > 160 * 1024) return CLLM_E_UNSUPPORTED;
    if (epi == 1 && (nrows % 2 || (nrows / 2) % 8 || bias || resid)) return CLLM_E_UNSUPPORTED;
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    int64_t grid = (units + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / kind);
    const size_t lds = act_row_bytes(K, kind) + 16 * (size_t)(wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);      // (the record area stays: the kernel's LDS layout is k_gemv_dec's)
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
    if (pro == 1 && npre == 8) return CLLM_E_UNSUPPORTED;
#define GOF3(FMT_, PRO_, EPI_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (lds > 64 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_dec<FMT_, PRO_, EPI_, NPRE_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_dec<FMT_, PRO_, EPI_, NPRE_, false, true>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const float *) nullptr, (const char *) W, nblk, kfull, nrem, eps, dst, (float *) nullptr, \
                           bias, resid, (unsigned long long *) nullptr, (const int32_t *) nullptr, 0ull, 0, 0); } while (0)
#define GOF(FMT_) do { \
        if (pro == 1 && epi == 1) { if (npre == 1) GOF3(FMT_, 1, 1, 1); else GOF3(FMT_, 1, 1, 4); } \
        else if (pro == 1)        { if (npre == 1) GOF3(FMT_, 1, 0, 1); else GOF3(FMT_, 1, 0, 4); } \
        else                      { if (npre == 1) GOF3(FMT_, 2, 0, 1); else if (npre == 4) GOF3(FMT_, 2, 0, 4); else GOF3(FMT_, 2, 0, 8); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GOF(CLLM_TYPE_Q4_K); else if (wtype == CLLM_TYPE_Q4_0) GOF(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOF(CLLM_TYPE_Q4_1); else GOF(CLLM_TYPE_Q8_0);
#undef GOF
#undef GOF3
    LAUNCH_CHECK();
    return CLLM_OK;
}
