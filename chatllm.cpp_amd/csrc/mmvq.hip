// mmvq.hip -- quantized mat-vec for 1..8 PRE-QUANTIZED activation columns, Q4_K / Q4_0 / Q8_0 weights (small-batch MUL_MAT,
// MUL_MAT_ID).  The single-column decode launches with the activation produced inside the kernel live in gemv_decode.hip.
//
// Replaces the decode branch of ggml_compute_forward_mul_mat (ggml/src/ggml-cpu/ggml-cpu.c:1359-1421) and its
// vec_dot kernels ggml_vec_dot_q4_K_q8_K / q4_0_q8_0 / q8_0_q8_0 (ggml-cpu/quants.c:550-623, 115-150, 305-333):
// integer block dot products (exact), fp32 scale + accumulate.
//
// HBM-bound by construction: every weight byte is read exactly once with 16-byte loads, the quantized
// activation row(s) live in LDS and are shared by all waves of the workgroup.  Algorithmic bytes per
// launch = nrows * row_size(type, K) (+ the activation row, negligible).
//
// Work split
//   Q4_K : a group of 8 lanes owns one 144-byte super-block per step: all 8 lanes fetch the 16-byte
//          header {d, dmin, 12 packed 6-bit scales/mins} (one broadcast request), lane j fetches qs[16j..16j+15]
//          (two 64-weight halves: low nibbles belong to sub-block 2*(j/2), high nibbles to 2*(j/2)+1) and
//          owns min #j.  A wave therefore streams 8 consecutive super-blocks = 1152 contiguous bytes per step.
//   Q4_0 / Q8_0 : one lane per 32-weight block (18 / 34 bytes, 2-byte aligned; gfx950 serves the
//          misaligned dwordx4 directly), a wave streams 64 consecutive blocks per step.
// One wave owns one weight row at a time; rows are dealt round-robin to the waves of the grid.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"
#include "q32.h"

static int g_mmvq_wg = 256;      // threads per workgroup (tunable: CLLM_MMVQ_WG)
static int g_mmvq_wgs_per_cu = 8; // grid cap (tunable: CLLM_MMVQ_OCC)

// kernel arguments; `ids` != NULL turns the launch into MUL_MAT_ID: blockIdx.y enumerates (slot u, token t) pairs,
// each with its own expert matrix (ggml-cpu.c:1432-1678).
struct mmvq_args {
    const char * W; int64_t nb01, nb02; int64_t nrows; int nblk;
    const char * act; size_t act_stride;
    float * dst; int64_t dst_cs;
    const int32_t * ids; int64_t ids_s0, ids_s1;   // strides in int32 units
    int n_used, b_ne1, n_as; int64_t dst_s1, dst_s2; // dst strides (floats) over (u, t)
    const float * bias;                            // optional fused ADD of a per-row bias   (Linear::forward, src/layers.cpp:2111-2129)
    const float * resid;                           // optional fused ADD of the residual     (LMBlock1Forward::forward :2740,:2758), indexed like dst
};

__device__ __forceinline__ bool mmvq_select(const mmvq_args & a, const char *& W, const char *& act, float *& dst) {
    W = a.W; act = a.act; dst = a.dst;
    if (a.ids) {
        const int u = blockIdx.y % a.n_used, t = blockIdx.y / a.n_used;
        const int e = a.ids[u * a.ids_s0 + t * a.ids_s1];
        if (e < 0 || e >= a.n_as) return false;      // the CPU asserts; we skip the column
        W   += (int64_t) e * a.nb02;
        act += ((int64_t)(u % a.b_ne1) + (int64_t) t * a.b_ne1) * a.act_stride;
        dst += u * a.dst_s1 + t * a.dst_s2;
    }
    return true;
}

// ---- LDS activation row(s): staged from global, or produced in place (fused RMS_NORM / quantization) -----------------
__device__ __forceinline__ void stage_act(char * lds, const char * __restrict__ act, size_t act_stride, size_t row_bytes, int nc) {
    const int n16 = (int)(row_bytes / 16);
    for (int c = 0; c < nc; c++) {
        const u32x4 * s = (const u32x4 *)(act + c * act_stride);
        u32x4 * d = (u32x4 *)(lds + c * row_bytes);
        for (int i = threadIdx.x; i < n16; i += blockDim.x) d[i] = s[i];
    }
}

// ---- Q4_K -----------------------------------------------------------------------------------------------
// The (row, step) pairs a wave owns form one linear sequence of steps (a step = 8 super-blocks = 1152 contiguous bytes);
// the loads of the first P steps are issued BEFORE the activation prologue and every consumed step immediately re-issues
// the load P steps ahead.  Decode launches use P = 2 with one 1024-thread workgroup per CU (measured: deeper prefetch is
// SLOWER on MI355X -- 4 steps -2 %, 8 steps -15 % tokens/s -- more requests in flight than the memory system wants).
// Accumulation order inside a row does not depend on P.
template <int NC, int P>
__global__ void __launch_bounds__(1024) k_mmvq_q4_K(const mmvq_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int EPI = 0, PP = P;
    const char * W = a.W; const char * act = a.act; float * dst = a.dst;
    if (!mmvq_select(a, W, act, dst)) return;
    const int64_t nb01 = a.nb01, nrows = a.nrows, dst_cs = a.dst_cs; const int nblk = a.nblk;
    const int64_t K = (int64_t) nblk * 256;
    const size_t  rb = act_row_bytes(K, 256);

    const int lane = threadIdx.x & 63;
    const int grp  = lane >> 3;            // which of the 8 super-blocks of this step
    const int j    = lane & 7;             // which 16-byte slice of qs / which min
    const int waves_per_wg = blockDim.x >> 6;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);                                           // scalar
    const int64_t nwaves = (int64_t) gridDim.x * waves_per_wg;
    const int S = (nblk + 7) >> 3;         // steps per row

    // A wave owns `units`: one row, or (epilogue 1) the gate / up row pair of one feature.  Units are dealt in rounds of
    // nwaves: in a full round wave (b, w) takes unit round*nwaves + 16 b + w (a workgroup streams 16 consecutive rows); the
    // last, partial round is dealt workgroup-interleaved (w * gridDim + b) so that every CU gets the same share of it.
    constexpr int RU = EPI == 1 ? 2 : 1;
    const int64_t nunits = nrows / RU;
    const int64_t kfull = nunits / nwaves, nrem = nunits - kfull * nwaves;
    const int64_t lin = (int64_t) blockIdx.x * waves_per_wg + wave_in_wg, alt = (int64_t) wave_in_wg * gridDim.x + blockIdx.x;
    const int64_t nmine = kfull + (alt < nrem ? 1 : 0);                    // units of this wave
    auto unit_of = [&](int64_t k) { return k * nwaves + (k < kfull ? lin : alt); };

    // Loads are unconditional (out-of-range steps re-read block 0 of row 0 and are masked when consumed) so that the loop
    // body is straight-line code and the compiler's s_waitcnt vmcnt(N) counts stay exact: a consumed step waits for its
    // own two loads only, not for the 2 (P - 1) younger ones.
    u32x4 hh[PP], qq[PP];
    int64_t ik = 0; int isub = 0, is = 0;      // issue cursor: (unit ordinal, row of the unit, step of the row)
    auto issue = [&](u32x4 & h, u32x4 & q) {
        const int b = 8 * is + grp;
        const bool ok = ik < nmine && b < nblk;
        const char * bp = ok ? W + (unit_of(ik) * RU + isub) * nb01 + (int64_t) b * 144 : W;
        // (non-temporal loads were measured 4 % SLOWER here: the header and the quants of a block share cache lines)
        h = *(const u32x4 *) bp;                        // d|dmin, scales[0..3], [4..7], [8..11] (one broadcast request per 8 lanes)
        q = *(const u32x4 *)(bp + 16 + 16 * j);
        if (++is == S) { is = 0; if (++isub == RU) { isub = 0; ik++; } }
    };
#pragma unroll
    for (int p = 0; p < PP; p++) issue(hh[p], qq[p]);          // weight loads fly while the activation rows are staged
    stage_act(lds, act, a.act_stride, rb, NC);
    __syncthreads();

    const q4k_sel L = q4k_lane_sel(lane);
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 256);
    char * chain = lds + NC * rb + (size_t) wave_in_wg * NC * Q4K_CHAIN_BYTES;      // this wave's chain records, one buffer per column
    const int l16 = lane & 15;

    float accd[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) accd[c] = 0.0f;
    int64_t ck = 0; int csub = 0, cs = 0;      // consume cursor
    float gate = 0.0f;
    int cold = PP - P;                         // consumed steps that were covered by the prologue burst: no re-issue for them
    while (ck < nmine) {
#pragma unroll
        for (int p = 0; p < PP; p++) {
            const int b = 8 * cs + grp;
            const bool ok = ck < nmine && b < nblk;
            const int bb = ok ? b : 0;     // in-range LDS addresses for masked steps
#pragma unroll
            for (int c = 0; c < NC; c++) q4k_emit(hh[p], qq[p], lds + c * rb, off_d, off_s, bb, ok, L, chain + c * Q4K_CHAIN_BYTES + (cs & 1) * (4 * Q4K_PAIR_BYTES));
            if (PP == P) issue(hh[p], qq[p]);
            else if (cold > 0) cold--;      // (scalar)
            else issue(hh[(p + P) % PP], qq[(p + P) % PP]);
            const bool row_done = cs + 1 == S;
            if ((cs & 1) || row_done) {     // the fp32 chains in the reference's AVX2 order (q4k.h): every 16 super-blocks and at the row's end
                wave_lds_fence();
#pragma unroll
                for (int c = 0; c < NC; c++) q4k_chain(chain + c * Q4K_CHAIN_BYTES, (cs & 1) ? 8 : 4, l16, accd[c]);
                wave_lds_fence();
            }
            if (++cs == S) {                // row complete: finish the chains, epilogue, store (lane 0)
                const int64_t cunit = unit_of(ck), crow = cunit * RU + csub;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    float v = chain_finish<1>(accd[c]);
                    if (ck < nmine) {   // wave-uniform; bias / resid come through the scalar cache (their own counter)
                        if (EPI == 1) {
                            if (csub == 0) gate = v;
                            else if (lane == 0) dst[cunit] = (gate / (1.0f + ggml_expf_poly(0.0f - gate))) * v;
                        } else {
                            if (a.bias)  v = v + uniform_load_f32(a.bias + crow);
                            if (a.resid) v = v + uniform_load_f32(a.resid + crow + c * dst_cs);
                            if (lane == 0) dst[crow + c * dst_cs] = v;
                        }
                    }
                    accd[c] = 0.0f;
                }
                cs = 0;
                if (++csub == RU) { csub = 0; ck++; }
            }
        }
    }
}

// ---- Q4_0 / Q8_0 (one lane per 32-weight block) ------------------------------------------------------------

template <int NC, int FMT>
__global__ void __launch_bounds__(1024) k_mmvq_q32(const mmvq_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const char * W; const char * act; float * dst;
    if (!mmvq_select(a, W, act, dst)) return;
    const int64_t nb01 = a.nb01, nrows = a.nrows, dst_cs = a.dst_cs; const int nblk = a.nblk;
    const int64_t K = (int64_t) nblk * 32;
    const size_t  rb = act_row_bytes(K, 32);
    stage_act(lds, act, a.act_stride, rb, NC);
    __syncthreads();

    constexpr int BS = q32_fmt<FMT>::BS;
    const int lane = threadIdx.x & 63;
    const int waves_per_wg = blockDim.x >> 6;
    const int64_t wave0 = (int64_t) blockIdx.x * waves_per_wg + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t nwaves = (int64_t) gridDim.x * waves_per_wg;

    char * chain = lds + NC * rb + (size_t)(threadIdx.x >> 6) * NC * Q32_CHAIN_BYTES;      // this wave's chain records, one buffer per column
    const int l16 = lane & 15;
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 32);
    for (int64_t row = wave0; row < nrows; row += nwaves) {
        const char * wr = W + row * nb01;
        float acc[NC];
#pragma unroll
        for (int c = 0; c < NC; c++) acc[c] = 0.0f;
        for (int b0 = 0; b0 < nblk; b0 += 64) {
            const int b = b0 + lane;
            const bool ok = b < nblk;
            uint32_t d16 = 0; u32x4 q0 = {0, 0, 0, 0}, q1 = {0, 0, 0, 0};
            if (ok) q32_load<FMT>(wr + (int64_t) b * BS, d16, q0, q1);
#pragma unroll
            for (int c = 0; c < NC; c++) q32_emit<FMT>(d16, q0, q1, lds + c * rb, off_d, off_s, ok ? b : 0, ok, lane, chain + c * Q32_CHAIN_BYTES);
            wave_lds_fence();               // the fp32 chains in the reference's AVX2 order (q32.h), one step (64 blocks) at a time
#pragma unroll
            for (int c = 0; c < NC; c++) q32_chain<FMT>(chain + c * Q32_CHAIN_BYTES, l16, acc[c]);
            wave_lds_fence();
        }
#pragma unroll
        for (int c = 0; c < NC; c++) {
            float v = chain_finish<q32_fmt<FMT>::IS_Q41 ? 2 : 0>(acc[c]);
            if (a.bias)  v = v + a.bias[row];
            if (a.resid) v = v + a.resid[row + c * dst_cs];
            if (lane == 0) dst[row + c * dst_cs] = v;
        }
    }
}

// ---- exact integer sums (tier T0 KAT) -------------------------------------------------------------------------
__global__ void k_isums(int wtype, int64_t K, const char * __restrict__ w, const char * __restrict__ act, int32_t * __restrict__ out) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    const int8_t * aq = (const int8_t *) act;
    if (wtype == CLLM_TYPE_Q4_K) {
        if (b >= K / 256) return;
        const block_q4_K * x = (const block_q4_K *) w + b;
        const int32_t * ss = (const int32_t *)(act + act_off_s(K, 256));
        int tot = 0, mins = 0;
        for (int s = 0; s < 8; s++) {
            int sc, m;
            if (s < 4) { sc = x->scales[s] & 63; m = x->scales[s + 4] & 63; }
            else { sc = (x->scales[s + 4] & 0xF) | ((x->scales[s - 4] >> 6) << 4); m = (x->scales[s + 4] >> 4) | ((x->scales[s] >> 6) << 4); }
            int dsum = 0;
            for (int l = 0; l < 32; l++) {
                const uint8_t qb = x->qs[(s >> 1) * 32 + l];
                const int q = (s & 1) ? (qb >> 4) : (qb & 0xF);
                dsum += q * aq[b * 256 + s * 32 + l];
            }
            tot += sc * dsum;
            mins += m * ss[b * 8 + s];
        }
        out[2 * b] = tot; out[2 * b + 1] = mins;
    } else {
        if (b >= K / 32) return;
        int s = 0;
        if (wtype == CLLM_TYPE_Q8_0) {
            const block_q8_0 * x = (const block_q8_0 *) w + b;
            for (int l = 0; l < 32; l++) s += x->qs[l] * aq[b * 32 + l];
        } else if (wtype == CLLM_TYPE_Q4_1) {
            const block_q4_1 * x = (const block_q4_1 *) w + b;
            for (int l = 0; l < 16; l++) { s += (x->qs[l] & 0xF) * aq[b * 32 + l]; s += (x->qs[l] >> 4) * aq[b * 32 + l + 16]; }
        } else {
            const block_q4_0 * x = (const block_q4_0 *) w + b;
            for (int l = 0; l < 16; l++) { s += ((x->qs[l] & 0xF) - 8) * aq[b * 32 + l]; s += ((x->qs[l] >> 4) - 8) * aq[b * 32 + l + 16]; }
        }
        out[b] = s;
    }
}

extern "C" int cllm_vec_dot_isums(void * stream, int wtype, int64_t k, const void * w_row, const float * x, int32_t * isums) {
    hipStream_t st = (hipStream_t) stream;
    const int kb = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (!is_quant_type(wtype)) FAIL(CLLM_E_UNSUPPORTED, "vec_dot_isums: type %d", wtype);
    if (k <= 0 || k % kb) FAIL(CLLM_E_INVALID, "vec_dot_isums: k");
    void * act = nullptr;
    const size_t bytes = act_row_bytes(k, kb);
    HIP_TRY(hipMalloc(&act, bytes));
    tview s; s.data = (char *) x; s.ne[0] = k; s.ne[1] = s.ne[2] = s.ne[3] = 1; s.nb[0] = 4; s.nb[1] = s.nb[2] = s.nb[3] = k * 4;
    int rc = launch_quantize_act(st, act_kind_of(wtype), s, act, bytes);
    if (rc == CLLM_OK) {
        const int64_t nb = k / kb;
        hipLaunchKernelGGL(k_isums, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, wtype, k, (const char *) w_row, (const char *) act, isums);
        rc = cllm_hip_check(hipGetLastError(), "k_isums", __FILE__, __LINE__);
    }
    (void) hipStreamSynchronize(st);
    (void) hipFree(act);
    return rc;
}

// ---- host launcher ---------------------------------------------------------------------------------------------
static void mmvq_tunables() {
    static bool done = false;
    if (done) return;
    done = true;
    if (const char * e = getenv("CLLM_MMVQ_WG"))  { int v = atoi(e); if (v == 64 || v == 128 || v == 256 || v == 512) g_mmvq_wg = v; }
    if (const char * e = getenv("CLLM_MMVQ_OCC")) { int v = atoi(e); if (v >= 1 && v <= 32) g_mmvq_wgs_per_cu = v; }
    if (const char * e = getenv("CLLM_MMVQ_WG"))   { int v = atoi(e); if (v == 1024) g_mmvq_wg = v; }
}

template <typename KernelT>
static int launch_one(hipStream_t st, KernelT kern, size_t lds_bytes, size_t chain_per_wave, const mmvq_args & a_in, int grid_y) {
    mmvq_args a = a_in;
    mmvq_tunables();
    const int wg = g_mmvq_wg, wpw = wg / 64;
    lds_bytes += (size_t) wpw * chain_per_wave;              // activation rows | the waves' chain records (q4k.h / q32.h)
    if (lds_bytes > 160 * 1024) FAIL(CLLM_E_UNSUPPORTED, "mmvq: %zu bytes of LDS", lds_bytes);
    int64_t grid = (a.nrows + wpw - 1) / wpw;
    int64_t cap = (int64_t) device_cu_count() * g_mmvq_wgs_per_cu / grid_y;
    if (cap < 1) cap = 1;
    if (grid > cap) grid = cap;     // (balancing rows per wave exactly was measured slower than simply using more workgroups)
    if (lds_bytes > 64 * 1024) HIP_TRY(hipFuncSetAttribute((const void *) kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, dim3((unsigned) grid, (unsigned) grid_y), dim3(wg), lds_bytes, st, a);
    LAUNCH_CHECK();
    return CLLM_OK;
}

static int mmvq_dispatch(hipStream_t st, int wtype, int nc, size_t lds, const mmvq_args & a, int grid_y) {
    const size_t chain = (size_t) nc * (wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);
#define GO(...) return launch_one(st, __VA_ARGS__, lds, chain, a, grid_y)
    if (wtype == CLLM_TYPE_Q4_K)      { if (nc == 4) GO(k_mmvq_q4_K<4, 1>); else if (nc == 2) GO(k_mmvq_q4_K<2, 1>); else GO(k_mmvq_q4_K<1, 1>); }
    else if (wtype == CLLM_TYPE_Q8_0) { if (nc == 4) GO(k_mmvq_q32<4, CLLM_TYPE_Q8_0>); else if (nc == 2) GO(k_mmvq_q32<2, CLLM_TYPE_Q8_0>); else GO(k_mmvq_q32<1, CLLM_TYPE_Q8_0>); }
    else if (wtype == CLLM_TYPE_Q4_0) { if (nc == 4) GO(k_mmvq_q32<4, CLLM_TYPE_Q4_0>); else if (nc == 2) GO(k_mmvq_q32<2, CLLM_TYPE_Q4_0>); else GO(k_mmvq_q32<1, CLLM_TYPE_Q4_0>); }
    else if (wtype == CLLM_TYPE_Q4_1) { if (nc == 4) GO(k_mmvq_q32<4, CLLM_TYPE_Q4_1>); else if (nc == 2) GO(k_mmvq_q32<2, CLLM_TYPE_Q4_1>); else GO(k_mmvq_q32<1, CLLM_TYPE_Q4_1>); }
#undef GO
    FAIL(CLLM_E_UNSUPPORTED, "mmvq: weight type %d", wtype);
}

// w: [K, nrows] quant rows (dims 2,3 handled by the caller), act: ncols rows of act layout, dst: column c at dst.data + c*nb1
int launch_mmvq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, int64_t ncols,
                const tview & /*src1_geom*/, const tview & dst) {
    const int64_t K = w.ne[0];
    const int kb = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    const size_t rb = act_row_bytes(K, kb);
    if (dst.nb[1] % 4) FAIL(CLLM_E_INVALID, "mmvq: dst stride");
    mmvq_args a = {};
    a.W = w.data; a.nb01 = w.nb[1]; a.nb02 = w.nb[2]; a.nrows = w.ne[1]; a.nblk = (int)(K / kb);
    a.act_stride = act_stride; a.dst_cs = dst.nb[1] / 4;
    int64_t c = 0;
    while (c < ncols) {
        int nc = ncols - c >= 4 ? 4 : (ncols - c >= 2 ? 2 : 1);
        mmvq_tunables();
        const size_t chw = (size_t)(g_mmvq_wg / 64) * (wtype == CLLM_TYPE_Q4_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES);
        while (nc > 1 && (size_t) nc * (rb + chw) > 160 * 1024) nc >>= 1;
        if ((size_t) nc * (rb + chw) > 160 * 1024) FAIL(CLLM_E_UNSUPPORTED, "mmvq: K=%lld does not fit LDS", (long long) K);
        a.act = (const char *) act + c * act_stride;
        a.dst = (float *)(dst.data + c * dst.nb[1]);
        const int rc = mmvq_dispatch(st, wtype, nc, (size_t) nc * rb, a, 1);
        if (rc) return rc;
        c += nc;
    }
    return CLLM_OK;
}

// MUL_MAT_ID for few tokens: one grid.y slice per (slot, token), expert picked on the device from ids
int launch_mmvq_id(hipStream_t st, int wtype, const tview & as, const void * act, size_t act_stride, int64_t b_ne1,
                   const tview & ids, const tview & dst) {
    const int64_t K = as.ne[0];
    const int kb = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    const size_t rb = act_row_bytes(K, kb);
    if (rb > 160 * 1024) FAIL(CLLM_E_UNSUPPORTED, "mmvq_id: K=%lld does not fit LDS", (long long) K);
    const int64_t n_used = ids.ne[0], n_tok = ids.ne[1];
    if (n_used * n_tok > 65535) FAIL(CLLM_E_UNSUPPORTED, "mmvq_id: too many (slot, token) pairs");
    mmvq_args a = {};
    a.W = as.data; a.nb01 = as.nb[1]; a.nb02 = as.nb[2]; a.nrows = as.ne[1]; a.nblk = (int)(K / kb);
    a.act = (const char *) act; a.act_stride = act_stride; a.dst = (float *) dst.data; a.dst_cs = 0;
    a.ids = (const int32_t *) ids.data; a.ids_s0 = ids.nb[0] / 4; a.ids_s1 = ids.nb[1] / 4;
    a.n_used = (int) n_used; a.b_ne1 = (int) b_ne1; a.n_as = (int) as.ne[2]; a.dst_s1 = dst.nb[1] / 4; a.dst_s2 = dst.nb[2] / 4;
    return mmvq_dispatch(st, wtype, 1, rb, a, (int)(n_used * n_tok));
}

// single pre-quantized activation row (decode): dst[row] = W[row] . act (+ bias[row]) (+ resid[row]); dst may alias resid
int launch_mmvq_act(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, const void * act, float * dst,
                    const float * bias, const float * resid) {
    const int kb = wtype == CLLM_TYPE_Q4_K ? 256 : 32;
    if (K % kb) FAIL(CLLM_E_INVALID, "mmvq_act: K");
    const size_t rb = act_row_bytes(K, kb);
    if (rb > 160 * 1024) FAIL(CLLM_E_UNSUPPORTED, "mmvq_act: K=%lld does not fit LDS", (long long) K);
    mmvq_args a = {};
    a.W = (const char *) W; a.nb01 = (int64_t) cllm_row_size(wtype, K); a.nb02 = 0; a.nrows = nrows; a.nblk = (int)(K / kb);
    a.act = (const char *) act; a.act_stride = rb; a.dst = dst; a.dst_cs = 0; a.bias = bias; a.resid = resid;
    return mmvq_dispatch(st, wtype, 1, rb, a, 1);
}

// decode mat-vec with the activation produced inside the kernel (gemv_decode.hip):
//   pro 1: act = quantize(rms_norm(px) * pw)     pro 2: act = quantize(px)     pro 3: act = quantize(silu(px[2i]) * px[2i+1]), i < K
//   epi 1: W rows alternate gate_u, up_u; dst[u] = silu(W[2u].act) * (W[2u+1].act), u < nrows / 2
int launch_mmvq_fused(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps,
                      int epi, float * dst, const float * bias, const float * resid) {
    const int rc = launch_gemv_decode(st, wtype, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid);
    if (rc == CLLM_E_UNSUPPORTED) cllm_set_error("mmvq_fused: type %d, K=%lld, prologue %d is outside the decode kernel's range", wtype, (long long) K, pro);
    return rc;
}
