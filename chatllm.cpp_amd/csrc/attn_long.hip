// attn_long.hip -- single-token attention for LONG contexts, split over the whole chip.
//
// k_attn_dec (decode_fused.hip) gives one workgroup per query head: 32 CUs of 256, each limited by what one CU can keep in
// flight (~55 GB/s), and the four query heads of a GQA group each re-read the same K / V rows: at 4096 cached positions the
// launch takes 40 us, at 16384 170 us.  Here the same arithmetic is cut into three launches that every CU takes part in:
//   k_attn_long_scores : grid (split, kv head): RoPE of the group's query heads and of the new k; K rows of one slice of
//                        positions are read ONCE and dotted with all r2 query heads of the group -> scores[head][i] (global)
//   k_attn_long_softmax: grid (head): soft_max over the head's score row, probabilities rounded to fp16 (in place)
//   k_attn_long_pv     : grid (row chunk, kv head): V^T rows of the chunk are read ONCE and dotted with the r2 probability rows
// Every score, every probability and every context element is produced by the same lane-group arithmetic in the same order as in
// k_attn_dec / the unfused MUL_MAT + SOFT_MAX nodes (only WHICH workgroup computes an element changes), so the results are
// bit-identical to theirs.  Used by the runner above CLLM_ATTN_LONG cached positions (default and minimum 512, the measured
// crossover is ~400): two extra launches (~9 us) buy 25 us at 4096 positions and 140 us at 16384.
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

static unsigned long long * g_long_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_attn_long_ts(unsigned long long * dev_buf) { g_long_ts = dev_buf; }   // tools only
#define TS(k) do { if (ts && threadIdx.x == 0) ts[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)

__device__ __forceinline__ int uniform_load_i32_(const int32_t * p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// ---- (1) RoPE + cache write + scores -------------------------------------------------------------------------------------------
template <int HD, int MODE, int R2>
__global__ void __launch_bounds__(256) k_attn_long_scores(const float * __restrict__ qkv, const int32_t * __restrict__ pos_dev, const float * __restrict__ rope_cs,
                                                          int nh, int nkv, float scale, uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache,
                                                          int ML, float * __restrict__ S, unsigned long long * ts) {
    TS(0);
    __shared__ float qs[R2 * HD];          // the group's query heads after RoPE, rounded to fp16 (src1 of K.Q)
    __shared__ float knew[HD], vnew[HD];   // the new k (after RoPE) and v, rounded to fp16 like the cache
    constexpr int half = HD / 2, G = HD / 8, RPW = 64 / G, off = MODE == 0 ? 1 : half, U = 4;
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KD = nkv * HD, QD = nh * HD;
    const int pos = uniform_load_i32_(pos_dev), n_kv = pos + 1;
    // slice of positions of this workgroup: rounded up to whole passes of 4 waves x RPW rows x U
    constexpr int PASS = 4 * RPW * U;
    const int chunk = (((n_kv + (int) gridDim.x - 1) / (int) gridDim.x + PASS - 1) / PASS) * PASS;
    const int i_lo = blockIdx.x * chunk, i_hi = min(n_kv, i_lo + chunk);
    if (i_lo >= n_kv) return;

    for (int t = tid; t < (R2 + 1) * half; t += 256) {          // pairs of the r2 query heads, then of k
        const int which = t / half, i = t - which * half, ic = MODE == 0 ? 2 * i : i;
        const float * x = which < R2 ? qkv + (g * R2 + which) * HD : qkv + QD + g * HD;
        const float x0 = x[ic], x1 = x[ic + off], c = rope_cs[2 * i], s_ = rope_cs[2 * i + 1];
        const float y0 = rope_rot_a(x0, x1, c, s_), y1 = rope_rot_b(x0, x1, c, s_);
        float * o = which < R2 ? qs + which * HD : knew;
        o[ic] = h2f(f2h(y0)); o[ic + off] = h2f(f2h(y1));
    }
    for (int d = tid; d < HD; d += 256) vnew[d] = h2f(f2h(qkv[QD + KD + g * HD + d]));
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int d = tid; d < HD; d += 256) {
            k_cache[(int64_t) pos * KD + g * HD + d] = f2h(knew[d]);
            v_cache[((int64_t) g * HD + d) * ML + pos] = f2h(vnew[d]);
        }
    }

    TS(1);
    // A wave owns 16 consecutive positions per pass (U = 4 rounds of RPW = 4 rows, 16 lanes per row: 256 contiguous bytes of the
    // row).  Each lane forms its 8-element partial per query head; the 16-lane reduction is NOT done with shuffles (4 heads x 4
    // ds_bpermute stages per row made this loop LDS-crossbar bound, 2.7 us per pass): the partials go through LDS once, and lane
    // (row, head) adds the row's 16 partials in the butterfly's order ((p0+p8)+(p4+p12))+... -> the same bits, 30x fewer LDS ops.
    static_assert(G == 16 || G == 8, "head sizes 128 / 64");
    constexpr int RPI = RPW * U;                                  // rows per wave and pass (16 for HD 128, 32 for HD 64)
    constexpr int PSTR = G * R2 + 4;                              // floats per row in the exchange buffer (+4: conflict-free column reads)
    __shared__ float part[4][RPI * PSTR];
    const int gl = lane & (G - 1), sub = lane / G;
    const int d = gl * 8;
    float q[R2][8];
#pragma unroll
    for (int h = 0; h < R2; h++)
#pragma unroll
        for (int j = 0; j < 8; j++) q[h][j] = qs[h * HD + d + j];
    const uint16_t * kbase = k_cache + g * HD + d;
    float * mypart = part[wave];
    auto load_rows = [&](int ib, u32x4 (&r)[U]) {                // unconditional (clamped): exact vmcnt bookkeeping
#pragma unroll
        for (int u = 0; u < U; u++) { const int i0 = ib + u * RPW + sub; r[u] = *(const u32x4 *)(kbase + (int64_t)(i0 < pos ? i0 : 0) * KD); }
    };
    u32x4 cur[U], nxt[U];
    constexpr int WSTEP = 4 * RPI;                                // positions per workgroup pass
    load_rows(i_lo + wave * RPI, cur);
    for (int ib = i_lo + wave * RPI; ib < i_hi; ib += WSTEP) {
        load_rows(ib + WSTEP, nxt);                              // the next pass is in flight while this one is consumed
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int rl = u * RPW + sub, i0 = ib + rl;
            float kv[8];
            if (i0 == pos) {
#pragma unroll
                for (int j = 0; j < 8; j++) kv[j] = knew[d + j];
            } else {
                const uint32_t wv[4] = { cur[u].x, cur[u].y, cur[u].z, cur[u].w };
#pragma unroll
                for (int j = 0; j < 4; j++) { kv[2*j] = h2f((uint16_t)(wv[j] & 0xffff)); kv[2*j + 1] = h2f((uint16_t)(wv[j] >> 16)); }
            }
#pragma unroll
            for (int h = 0; h < R2; h++) {
                float acc = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) acc = __builtin_fmaf(kv[j], q[h][j], acc);
                mypart[rl * PSTR + gl * R2 + h] = acc;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // partials written by the other lanes of this wave
        for (int t = lane; t < RPI * R2; t += 64) {              // lane t = (row, head)
            const int rl = t / R2, h = t - rl * R2, i0 = ib + rl;
            const float * pp = mypart + rl * PSTR + h;
            float p[G];
#pragma unroll
            for (int l = 0; l < G; l++) p[l] = pp[l * R2];
            float r;
            if constexpr (G == 16) {
                const float a0 = p[0] + p[8], a1 = p[1] + p[9], a2 = p[2] + p[10], a3 = p[3] + p[11], a4 = p[4] + p[12], a5 = p[5] + p[13], a6 = p[6] + p[14], a7 = p[7] + p[15];
                const float b0 = a0 + a4, b1 = a1 + a5, b2 = a2 + a6, b3 = a3 + a7;
                r = (b0 + b2) + (b1 + b3);
            } else {
                const float b0 = p[0] + p[4], b1 = p[1] + p[5], b2 = p[2] + p[6], b3 = p[3] + p[7];
                r = (b0 + b2) + (b1 + b3);
            }
            if (i0 < i_hi) S[(int64_t)(g * R2 + h) * ML + i0] = r * scale;          // the SCALE node
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // the exchange buffer is reused by the next pass
#pragma unroll
        for (int u = 0; u < U; u++) cur[u] = nxt[u];
    }
    TS(2);
}

// ---- (2) soft_max over one head's scores; probabilities rounded to fp16 (as src1 of V.P) ------------------------------------
__global__ void __launch_bounds__(1024) k_attn_long_softmax(const int32_t * __restrict__ pos_dev, int ML, const float * __restrict__ S, uint16_t * __restrict__ P16) {
    extern __shared__ __attribute__((aligned(16))) float sm[];    // [n_kv] scores | [n_kv / 8] group sums
    __shared__ double red_d[1];
    __shared__ float  red_f[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_kv = uniform_load_i32_(pos_dev) + 1, nv = n_kv & ~7;
    const float * row = S + (int64_t) blockIdx.x * ML;
    uint16_t * prow = P16 + (int64_t) blockIdx.x * ML;
    float * sc = sm; float * gsum = sm + ML;
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += 1024) { const float v = row[i]; sc[i] = v; mx = fmaxf(mx, v); }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red_f[w]);
    // every wave exponentiates; the per-group float sums are then accumulated by ONE wave in k_soft_max's order
    // (lane l owns groups l, l + 64, ...; double accumulation; DPP tree) -> the node kernel's bits
    for (int gi = tid * 8; gi < nv; gi += 1024 * 8) {
        float e[8];
#pragma unroll
        for (int l = 0; l < 8; l++) { e[l] = ggml_expf_poly(sc[gi + l] - mx); sc[gi + l] = e[l]; }
        const float a0 = e[0] + e[4], a1 = e[1] + e[5], a2 = e[2] + e[6], a3 = e[3] + e[7];
        gsum[gi >> 3] = (a0 + a2) + (a1 + a3);
    }
    __syncthreads();
    if (wave == 0) {
        double sum = 0.0;
        for (int gq = lane; gq < (nv >> 3); gq += 64) sum += (double) gsum[gq];
        if (lane == 0) for (int i = nv; i < n_kv; i++) { const float e = libm_expf(sc[i] - mx); sc[i] = e; sum += (double) e; }
        sum = wave_sum_d(sum);
        if (lane == 0) red_d[0] = sum;
    }
    __syncthreads();
    const float inv = (float)(1.0 / red_d[0]);
    for (int i = tid; i < n_kv; i += 1024) prow[i] = f2h(sc[i] * inv);          // the fp16 rounding src1 of V.P gets (exact in fp32 later)
}

// ---- (3) ctx = V . P ------------------------------------------------------------------------------------------------------------
// n_kv > 512 here, so a whole wave owns one V^T row (launch_T() in matmul_f.hip: G = 64): tail elements first, then 16-byte chunks
// in increasing i, butterfly reduction.  One V chunk feeds the r2 heads of the group.  The probabilities (exactly representable in
// fp16: they were rounded to it) are staged through LDS as fp16, one segment of SEG positions at a time, so that the inner loop's
// only global loads are the V chunks (with the r2 x 32-byte P loads from L2 on its critical path this kernel took 44 us at 16K).
template <int HD, int R2, int DR>
__global__ void __launch_bounds__(256) k_attn_long_pv(const int32_t * __restrict__ pos_dev, int nh, int nkv, const uint16_t * __restrict__ v_cache, int ML,
                                                      const uint16_t * __restrict__ P, float * __restrict__ att) {
    constexpr int SEG = 8192, RPWV = DR / 4, UC = 4;              // positions per staged segment, rows per wave, V chunks in flight
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[]; // [R2][SEG] fp16 probabilities
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_kv = uniform_load_i32_(pos_dev) + 1, n8 = n_kv & ~7;
    const int it = n8 + lane, iv = lane * 8;
    float acc[RPWV][R2];
    const uint16_t * vrow[RPWV];
#pragma unroll
    for (int rr = 0; rr < RPWV; rr++) {
        const int d0 = blockIdx.x * DR + wave + 4 * rr;
        vrow[rr] = v_cache + ((int64_t) g * HD + (d0 < HD ? d0 : 0)) * ML;
        const float vt = it < n_kv ? h2f(vrow[rr][it]) : 0.0f;
#pragma unroll
        for (int h = 0; h < R2; h++) acc[rr][h] = it < n_kv ? __builtin_fmaf(vt, h2f(P[(int64_t)(g * R2 + h) * ML + it]), 0.0f) : 0.0f;
    }
    for (int seg0 = 0; seg0 < n8; seg0 += SEG) {
        const int seg1 = min(n8, seg0 + SEG);
        __syncthreads();                                         // the previous segment has been consumed
        constexpr int NT = R2 * (SEG / 8) / 256;                 // 16-byte copy tasks per thread; all loads of a batch are in flight together
        constexpr int NB = NT < 8 ? NT : 8;
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += NB) {
            u32x4 buf[NB];
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const int c = tid + 256 * (t0 + t), h = c / (SEG / 8), i = seg0 + (c - h * (SEG / 8)) * 8;
                buf[t] = *(const u32x4 *)(P + (int64_t)(g * R2 + h) * ML + (i < seg1 ? i : 0));
            }
#pragma unroll
            for (int t = 0; t < NB; t++) {
                const int c = tid + 256 * (t0 + t), h = c / (SEG / 8), il = (c - h * (SEG / 8)) * 8;
                *(u32x4 *)(psm + h * SEG + il) = buf[t];
            }
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RPWV; rr++) {
            const uint16_t * vr = vrow[rr];
            u32x4 ring[UC];
#pragma unroll
            for (int c = 0; c < UC; c++) { const int ic = seg0 + iv + 512 * c; ring[c] = *(const u32x4 *)(vr + (ic < seg1 ? ic : 0)); }
            for (int i0 = seg0 + iv; i0 < seg1; i0 += 512 * UC) {
#pragma unroll
                for (int c = 0; c < UC; c++) {
                    const int i = i0 + 512 * c;
                    const u32x4 cur = ring[c];
                    { const int inx = i + 512 * UC; ring[c] = *(const u32x4 *)(vr + (inx < seg1 ? inx : 0)); }    // unconditional (clamped) refill
                    if (i < seg1) {
                        // both operands are fp16 in registers: fma((float) v, (float) p, acc) is one v_fma_mix_f32 per element
                        // (the conversions are exact, so this is the cvt + fma of the other kernels bit for bit)
                        const half8 v = __builtin_bit_cast(half8, cur);
#pragma unroll
                        for (int h = 0; h < R2; h++) {
                            const half8 pq = __builtin_bit_cast(half8, *(const u32x4 *)(psm + h * SEG + (i - seg0)));
#pragma unroll
                            for (int j = 0; j < 8; j++) acc[rr][h] = __builtin_fmaf((float) v[j], (float) pq[j], acc[rr][h]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int rr = 0; rr < RPWV; rr++) {
        const int d0 = blockIdx.x * DR + wave + 4 * rr;
#pragma unroll
        for (int h = 0; h < R2; h++) {
            float r = acc[rr][h];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
            if (lane == 0 && d0 < HD) att[(g * R2 + h) * HD + d0] = r;
        }
    }
}

// CLLM_E_UNSUPPORTED -> the caller uses the single-launch kernels.  S: scratch of nh * ML * 6 bytes (fp32 scores + fp16 probabilities).
int launch_attn_long(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode,
                     uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * S, float * att) {
    const int r2 = nkv > 0 ? nh / nkv : 0;
    if ((hd != 64 && hd != 128) || nh % nkv || (r2 != 1 && r2 != 2 && r2 != 4 && r2 != 8) || ML % 8 || ML > (1 << 30)) return CLLM_E_UNSUPPORTED;
    const size_t lds = (size_t)(ML + ML / 8) * 4;
    if (lds > 150 * 1024) return CLLM_E_UNSUPPORTED;
    const float scale = 1.0f / sqrtf((float) hd);
    uint16_t * P16 = (uint16_t *)(S + (size_t) nh * ML);          // fp16 probabilities behind the fp32 scores
    const int cus = device_cu_count();
    int nsplit = cus / nkv; if (nsplit < 1) nsplit = 1; if (nsplit > 64) nsplit = 64;
    constexpr int DR = 8;
#define SC(HD_, MODE_, R2_) hipLaunchKernelGGL((k_attn_long_scores<HD_, MODE_, R2_>), dim3(nsplit, nkv), dim3(256), 0, st, qkv, pos_dev, rope_cs, nh, nkv, scale, k_cache, v_cache, (int) ML, S, g_long_ts)
#define SC2(HD_, R2_) do { if (mode == 0) SC(HD_, 0, R2_); else SC(HD_, 2, R2_); } while (0)
#define SC3(HD_) do { if (r2 == 1) SC2(HD_, 1); else if (r2 == 2) SC2(HD_, 2); else if (r2 == 4) SC2(HD_, 4); else SC2(HD_, 8); } while (0)
    if (hd == 128) SC3(128); else SC3(64);
    LAUNCH_CHECK();
    static bool attr = false;
    if (lds > 48 * 1024 && !attr) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_softmax, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); attr = true; }
    hipLaunchKernelGGL(k_attn_long_softmax, dim3(nh), dim3(1024), lds, st, pos_dev, (int) ML, (const float *) S, P16);
    LAUNCH_CHECK();
#define PV(HD_, R2_) hipLaunchKernelGGL((k_attn_long_pv<HD_, R2_, DR>), dim3(HD_ / DR, nkv), dim3(256), (size_t) R2_ * 8192 * 2, st, pos_dev, nh, nkv, (const uint16_t *) v_cache, (int) ML, (const uint16_t *) P16, att)
#define PV2(HD_) do { if (r2 == 1) PV(HD_, 1); else if (r2 == 2) PV(HD_, 2); else if (r2 == 4) PV(HD_, 4); else PV(HD_, 8); } while (0)
    static bool attr_pv = false;
    if (r2 == 8 && !attr_pv) {      // 8 x 16 KB of fp16 probabilities
        HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_pv<128, 8, DR>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 * 2));
        HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_pv<64, 8, DR>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 8192 * 2));
        attr_pv = true;
    }
    if (hd == 128) PV2(128); else PV2(64);
    LAUNCH_CHECK();
#undef SC
#undef SC2
#undef SC3
#undef PV
#undef PV2
    return CLLM_OK;
}
