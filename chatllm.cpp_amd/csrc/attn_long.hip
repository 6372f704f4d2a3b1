// attn_long.hip -- single-token attention for LONG contexts, split over the whole chip.
//
// k_attn_dec (decode_fused.hip) gives one workgroup per query head: 32 CUs of 256, each limited by what one CU can keep in
// flight (~55 GB/s), and the four query heads of a GQA group each re-read the same K / V rows: at 4096 cached positions the
// launch takes 40 us, at 16384 170 us.  Here the same arithmetic is cut into three launches that every CU takes part in:
//   k_attn_long_scores : grid (split, kv head): RoPE of the group's query heads and of the new k; K rows of one slice of
//                        positions are read ONCE and dotted with all r2 query heads of the group -> scores[head][i] (global)
//   k_attn_long_softmax: grid (head): soft_max over the head's score row, probabilities rounded to fp16 (in place)
//   k_attn_long_pv     : grid (row chunk, kv head, query head of the group): 16 V^T rows x the head's probability row
// Every score and every context element is accumulated in the ORDER of the reference's ggml_vec_dot_f16 (vec.cpp:264-, AVX2 + F16C:
// 32 fp32 accumulators, accumulator a takes elements a, a + 32, ... one fma each, GGML_F32x8_REDUCE's tree, the n mod 32 leftovers one by
// one in double) with the lane layout of k_attn_dec: 16 lanes per row, lane c carries accumulators 2c and 2c + 1.  The soft_max is
// k_soft_max's.  Only WHICH workgroup computes an element differs from k_attn_dec / the unfused MUL_MAT + SOFT_MAX nodes, so the
// results are bit-identical to theirs -- and to libggml-cpu.so -- at every context length.  Used by the runner above CLLM_ATTN_LONG
// cached positions (default and minimum 512).
#include "common.h"

#include "q4k.h"
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

// GGML_F32x8_REDUCE over the 32 accumulators held two per lane by 16 lanes (decode_fused.hip's vd32_reduce: the same definition)
__device__ __forceinline__ float al_xor4(float v) { return __int_as_float(lane_xor4_i(__float_as_int(v))); }
__device__ __forceinline__ float al_vd32_reduce(float a0, float a1) {
    a0 = a0 + dpp_f<DPP_ROW_ROR8>(a0); a1 = a1 + dpp_f<DPP_ROW_ROR8>(a1);
    a0 = a0 + al_xor4(a0);             a1 = a1 + al_xor4(a1);
    a0 = a0 + dpp_f<DPP_QUAD_XOR2>(a0); a1 = a1 + dpp_f<DPP_QUAD_XOR2>(a1);
    const float u = a0 + a1;
    return u + dpp_f<DPP_QUAD_XOR1>(u);
}

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
    constexpr int half = HD / 2, off = MODE == 0 ? 1 : half, U = 4;
    const int g = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KD = nkv * HD, QD = nh * HD;
    const int pos = uniform_load_i32_(pos_dev), n_kv = pos + 1;
    // slice of positions of this workgroup: rounded up to whole passes of 4 waves x 4 rows x U
    constexpr int PASS = 4 * 4 * U;
    const int chunk = (((n_kv + (int) gridDim.x - 1) / (int) gridDim.x + PASS - 1) / PASS) * PASS;
    const int i_lo = blockIdx.x * chunk, i_hi = min(n_kv, i_lo + chunk);
    if (i_lo >= n_kv) return;

    // the first pass of K rows depends only on the position: requested before the RoPE work (its latency overlaps the projections' loads and the rotation)
    constexpr int NCH = HD / 32, RPI = 4 * U;
    const int c16 = lane & 15, sub = lane >> 4;
    const uint16_t * kbase = k_cache + g * HD + 2 * c16;
    auto load_rows = [&](int ib, uint32_t (&r)[U][NCH]) {        // unconditional (clamped): exact vmcnt bookkeeping
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i0 = ib + u * 4 + sub;
            const uint16_t * kr = kbase + (int64_t)(i0 < pos ? i0 : 0) * KD;
#pragma unroll
            for (int i = 0; i < NCH; i++) r[u][i] = *(const uint32_t *)(kr + 32 * i);
        }
    };
    uint32_t cur[U][NCH], nxt[U][NCH];
    constexpr int WSTEP = 4 * RPI;                                // positions per workgroup pass
    load_rows(i_lo + wave * RPI, cur);

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
    // A wave owns RPI consecutive positions per pass (U rounds of 4 rows, 16 lanes per row).  Lane c of a row carries the accumulators 2c and 2c + 1 of
    // ggml_vec_dot_f16: one dword (two fp16) of the row per 32-element chunk, one fma per accumulator and chunk, then the reference's reduction tree (DPP).
    float qa[R2][NCH], qb[R2][NCH];
#pragma unroll
    for (int h = 0; h < R2; h++)
#pragma unroll
        for (int i = 0; i < NCH; i++) { qa[h][i] = qs[h * HD + 32 * i + 2 * c16]; qb[h][i] = qs[h * HD + 32 * i + 2 * c16 + 1]; }
    float kna[NCH], knb[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) { kna[i] = knew[32 * i + 2 * c16]; knb[i] = knew[32 * i + 2 * c16 + 1]; }
    for (int ib = i_lo + wave * RPI; ib < i_hi; ib += WSTEP) {
        load_rows(ib + WSTEP, nxt);                              // the next pass is in flight while this one is consumed
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i0 = ib + u * 4 + sub;
            float k0[NCH], k1[NCH];
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                k0[i] = i0 == pos ? kna[i] : h2f((uint16_t)(cur[u][i] & 0xffff));
                k1[i] = i0 == pos ? knb[i] : h2f((uint16_t)(cur[u][i] >> 16));
            }
#pragma unroll
            for (int h = 0; h < R2; h++) {
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int i = 0; i < NCH; i++) { a0 = __builtin_fmaf(k0[i], qa[h][i], a0); a1 = __builtin_fmaf(k1[i], qb[h][i], a1); }
                const float v = al_vd32_reduce(a0, a1);
                if (c16 == 0 && i0 < i_hi) S[(int64_t)(g * R2 + h) * ML + i0] = v * scale;          // the SCALE node
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int i = 0; i < NCH; i++) cur[u][i] = nxt[u][i];
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
        double rinv = 1.0 / sum;                                      // (ORDER: soft_total_order_safe, common.h)
        if (__builtin_expect(!soft_total_order_safe(rinv, n_kv >> 3), 0)) { wave_lds_fence(); rinv = 1.0 / soft_sum_serial_groups(gsum, nv >> 3, sc, nv, n_kv); }
        if (lane == 0) red_d[0] = rinv;
    }
    __syncthreads();
    const float inv = (float) red_d[0];
    for (int i = tid; i < n_kv; i += 1024) prow[i] = f2h(sc[i] * inv);          // the fp16 rounding src1 of V.P gets (exact in fp32 later)
}

// ---- (3) ctx = V . P ------------------------------------------------------------------------------------------------------------
// ggml_vec_dot_f16 over the cached positions: accumulator a takes positions a, a + 32, ... in order, so a V^T row is 32 serial fma chains of
// n_kv / 32 steps -- 16 lanes per row, two chains per lane (one dword of the row per 32-position chunk), the probability pairs from LDS (fp16: they
// were rounded to it, the conversions are exact).  A workgroup is 16 rows x ONE query head (the r2 heads of a group re-read the V rows through L2:
// the chains, not the bytes, bound this launch); the n_kv mod 32 leftovers are added one by one in double (products rounded to fp32 first).
template <int HD>
__global__ void __launch_bounds__(256) k_attn_long_pv(const int32_t * __restrict__ pos_dev, int nh, int nkv, const uint16_t * __restrict__ v_cache, int ML,
                                                      const uint16_t * __restrict__ P, float * __restrict__ att) {
    extern __shared__ __attribute__((aligned(16))) uint16_t psm[]; // [n_kv rounded up to 8] fp16 probabilities of this head
    __shared__ float tail[16][32];
    const int g = blockIdx.y, r2 = gridDim.z, h = g * r2 + blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_kv = uniform_load_i32_(pos_dev) + 1, np = n_kv & ~31, nch = np >> 5, ntail = n_kv - np;
    const int c16 = lane & 15, sub = lane >> 4, rl = wave * 4 + sub, d0 = blockIdx.x * 16 + rl;
    const uint16_t * vr = v_cache + ((int64_t) g * HD + d0) * ML + 2 * c16;
    constexpr int UC = 16;                                        // chunks in flight per lane
    uint32_t ring[UC];
#pragma unroll
    for (int c = 0; c < UC; c++) ring[c] = *(const uint32_t *)(vr + 32 * (c < nch ? c : 0));
    const uint16_t t0 = vr[np + 0 < n_kv - 2 * c16 ? np : 0], t1 = vr[np + 1 < n_kv - 2 * c16 ? np + 1 : 0];      // leftovers np + 2 c16 (+ 1), clamped
    const uint16_t * prow = P + (int64_t) h * ML;
    for (int i = tid * 8; i < n_kv; i += 256 * 8) *(u32x4 *)(psm + i) = *(const u32x4 *)(prow + i);      // (ML % 8 == 0: whole 16-byte chunks are inside the row)
    __syncthreads();
    float a0 = 0.0f, a1 = 0.0f;
    // both operands are fp16 in registers: fma((float) v, (float) p, acc) is one v_fma_mix_f32 (the conversions are exact: the cvt + fma of the other kernels bit for bit)
    const uint32_t * pp = (const uint32_t *) psm + c16;
    const uint32_t * vp = (const uint32_t *) vr;
    auto step = [&](uint32_t vw, uint32_t pw) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a0) : "v"(vw), "v"(pw));      // low halves
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a1) : "v"(vw), "v"(pw));      // high halves
    };
    int i0 = 0;
    for (; i0 + 2 * UC <= nch; i0 += UC) {                        // whole groups whose refills are inside the row: no clamps, immediate offsets
#pragma unroll
        for (int c = 0; c < UC; c++) {
            const uint32_t cur = ring[c];
            ring[c] = vp[16 * (i0 + UC + c)];
            step(cur, pp[16 * (i0 + c)]);
        }
    }
    for (; i0 < nch; i0 += UC) {
#pragma unroll
        for (int c = 0; c < UC; c++) {
            const int i = i0 + c;
            const uint32_t cur = ring[c];
            { const int inx = i + UC; ring[c] = vp[16 * (inx < nch ? inx : 0)]; }      // unconditional (clamped) refill
            if (i < nch) step(cur, pp[16 * i]);
        }
    }
    const float res = al_vd32_reduce(a0, a1);
    if (2 * c16 < ntail)     tail[rl][2 * c16]     = h2f(t0) * h2f(psm[np + 2 * c16]);
    if (2 * c16 + 1 < ntail) tail[rl][2 * c16 + 1] = h2f(t1) * h2f(psm[np + 2 * c16 + 1]);
    wave_lds_fence();
    if (c16 == 0) {
        double sacc = (double) res;
        for (int t = 0; t < ntail; t++) sacc += (double) tail[rl][t];
        att[h * HD + d0] = (float) sacc;
    }
}

// ---- (2 + 3) soft_max INSIDE the V.P launch: the contexts whose score row fits the LDS next to the V ring (ML <= AL_FUSED_MAX_ML) take two launches instead of three ----
// Every workgroup of a head (HD / 16 of them) redoes the head's soft_max with all of its 16 waves -- max, exponentials, group sums, the double sum by one wave in k_soft_max's order, fp16
// probabilities: the same bits in each -- while the first V tiles of its four V.P waves are already in flight: a launch less (~4 us of fixed cost + a boundary) for ~3 us of redundant
// exponentials.  The V^T rows travel HBM -> LDS by DMA (global_load_lds_dwordx4) into a per-wave ring of NS tiles of 4 rows x 512 positions: no VGPRs, whole tiles in
// flight (the dword-per-lane loads of the chain layout kept too few bytes in flight).  The chain is the one of k_attn_long_pv: 16 lanes per row, two accumulators per
// lane, one 32-position chunk per step from the ring (V) and the LDS probability row (P), v_fma_mix_f32 on the fp16 pairs.
#define AL_TILE 512
#define AL_NS 4
#define AL_FUSED_MAX_ML 12288
__device__ __forceinline__ void al_dma16(const char * base /* wave-uniform */, unsigned voff, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void al_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int HD>
__global__ void __launch_bounds__(1024) k_attn_long_softmax_pv(const int32_t * __restrict__ pos_dev, int nh, int nkv, const uint16_t * __restrict__ v_cache, int ML,
                                                              const float * __restrict__ S, float * __restrict__ att) {
    extern __shared__ __attribute__((aligned(16))) char lds[];    // [ML] fp32 exponentials | [ML / 8] group sums | [ML] fp16 probabilities | 4 waves x NS x 4 KB of V
    __shared__ double red_d[1];
    __shared__ float  red_f[16];
    __shared__ float tail[16][32];
    // 16 waves: all of them do the soft_max; waves 0..3 own the V ring and the V.P chains (4 rows each), the others leave after the soft_max
    const int g = blockIdx.y, r2 = gridDim.z, h = g * r2 + blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool pv_wave = wave < 4;
    const int n_kv = uniform_load_i32_(pos_dev) + 1, nv = n_kv & ~7, np = n_kv & ~31, nch = np >> 5, ntail = n_kv - np;
    const int c16 = lane & 15, sub = lane >> 4, rl = wave * 4 + sub, d0 = blockIdx.x * 16 + rl;
    float * ex = (float *) lds; float * gsum = ex + ML; uint16_t * p16 = (uint16_t *)(gsum + ML / 8);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *) lds;
    const unsigned ring = lds0 + (unsigned) ML * 4u + (unsigned)(ML / 8) * 4u + (unsigned) ML * 2u + (unsigned)(wave & 3) * (AL_NS * 4096u);

    // ---- the V ring: tile t = positions 512 t .. 512 t + 511 of this wave's 4 rows; lane l moves 16 bytes (8 positions) per row ----
    const int ntiles = (nch + 15) >> 4;
    const char * vrow0 = (const char *)(v_cache + ((int64_t) g * HD + blockIdx.x * 16 + (wave & 3) * 4) * ML);
    auto issue = [&](int t) {
        if (pv_wave && t < ntiles) {
            int p0 = t * AL_TILE + lane * 8;
            if (p0 > ML - 8) p0 = ML - 8;                           // (a ragged last tile stays inside the row: those positions are not used)
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring + (unsigned)(t % AL_NS) * 4096u);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's previous occupant has been read
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const char * rb = vrow0 + (size_t) r * (size_t) ML * 2;
                const char * rbs = (const char *)(((unsigned long long)(unsigned) __builtin_amdgcn_readfirstlane((int)((unsigned long long) rb >> 32)) << 32) |
                                                  (unsigned) __builtin_amdgcn_readfirstlane((int)(unsigned long long) rb));
                al_dma16(rbs, (unsigned) p0 * 2u, dst + 1024u * r);
            }
        }
    };
#pragma unroll
    for (int t = 0; t < AL_NS - 1; t++) issue(t);

    // ---- soft_max of head h (k_attn_long_softmax's arithmetic; 256 threads instead of 1024: the partition of the sums is the same) ----
    const float * row = S + (int64_t) h * ML;
    float mx = -INFINITY;
    for (int i = tid * 4; i < n_kv; i += 4096) {                    // (ML % 8 == 0: a 16-byte load stays inside the row; the tail past n_kv is masked)
        const f32x4 v = *(const f32x4 *)(row + i);
        *(f32x4 *)(ex + i) = v;
        mx = fmaxf(mx, v.x); if (i + 1 < n_kv) mx = fmaxf(mx, v.y); if (i + 2 < n_kv) mx = fmaxf(mx, v.z); if (i + 3 < n_kv) mx = fmaxf(mx, v.w);
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red_f[w]);
    for (int gi = tid * 8; gi < nv; gi += 1024 * 8) {
        const f32x4 s0 = *(const f32x4 *)(ex + gi), s1 = *(const f32x4 *)(ex + gi + 4);      // (16-byte LDS accesses: the 32-byte lane stride is 8-way conflicted for dwords)
        float e[8] = { ggml_expf_poly(s0.x - mx), ggml_expf_poly(s0.y - mx), ggml_expf_poly(s0.z - mx), ggml_expf_poly(s0.w - mx),
                       ggml_expf_poly(s1.x - mx), ggml_expf_poly(s1.y - mx), ggml_expf_poly(s1.z - mx), ggml_expf_poly(s1.w - mx) };
        *(f32x4 *)(ex + gi) = f32x4{e[0], e[1], e[2], e[3]}; *(f32x4 *)(ex + gi + 4) = f32x4{e[4], e[5], e[6], e[7]};
        const float a0 = e[0] + e[4], a1 = e[1] + e[5], a2 = e[2] + e[6], a3 = e[3] + e[7];
        gsum[gi >> 3] = (a0 + a2) + (a1 + a3);
    }
    __syncthreads();
    if (wave == 0) {
        double sum = 0.0;
        for (int gq = lane; gq < (nv >> 3); gq += 64) sum += (double) gsum[gq];
        if (lane == 0) for (int i = nv; i < n_kv; i++) { const float e = libm_expf(ex[i] - mx); ex[i] = e; sum += (double) e; }
        sum = wave_sum_d(sum);
        double rinv = 1.0 / sum;                                      // (ORDER: soft_total_order_safe, common.h)
        if (__builtin_expect(!soft_total_order_safe(rinv, n_kv >> 3), 0)) { wave_lds_fence(); rinv = 1.0 / soft_sum_serial_groups(gsum, nv >> 3, ex, nv, n_kv); }
        if (lane == 0) red_d[0] = rinv;
    }
    __syncthreads();
    const float inv = (float) red_d[0];
    for (int i = tid * 2; i < n_kv; i += 2048) {                    // the fp16 rounding src1 of V.P gets (exact in fp32 later)
        const uint32_t lo = f2h(ex[i] * inv), hi = i + 1 < n_kv ? f2h(ex[i + 1] * inv) : 0u;
        *(uint32_t *)(p16 + i) = lo | (hi << 16);
    }
    __syncthreads();
    if (!pv_wave) return;

    // ---- V.P: 32 serial chains per row, chunk by chunk ----
    float a0 = 0.0f, a1 = 0.0f;
    auto step = [&](uint32_t vw, uint32_t pw) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,1,0]" : "+v"(a0) : "v"(vw), "v"(pw));      // low halves
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "+v"(a1) : "v"(vw), "v"(pw));      // high halves
    };
    typedef __attribute__((address_space(3))) const uint32_t * lptr;
    const uint32_t * pp = (const uint32_t *) p16 + c16;
    for (int t = 0; t < ntiles; t++) {
        issue(t + AL_NS - 1);
        const int later = (t + AL_NS - 1 < ntiles ? AL_NS - 1 : ntiles - 1 - t);      // tiles issued after tile t: four DMA instructions each
        if (later >= 3) al_wait_vm<12>(); else if (later == 2) al_wait_vm<8>(); else if (later == 1) al_wait_vm<4>(); else al_wait_vm<0>();
        lptr vp = (lptr)(size_t)(ring + (unsigned)(t % AL_NS) * 4096u + (unsigned) sub * 1024u + (unsigned) c16 * 4u);
        const uint32_t * pt = pp + 256 * t;
        const int nc = nch - 16 * t < 16 ? nch - 16 * t : 16;
        if (nc == 16) {
#pragma unroll
            for (int c = 0; c < 16; c++) step(vp[16 * c], pt[16 * c]);
        } else {
            for (int c = 0; c < nc; c++) step(vp[16 * c], pt[16 * c]);
        }
    }
    const float res = al_vd32_reduce(a0, a1);
    if (ntail) {                                                     // the n_kv mod 32 leftovers, one by one in double (products rounded to fp32 first)
        const uint16_t * vr = v_cache + ((int64_t) g * HD + d0) * ML;
        if (2 * c16 < ntail)     tail[rl][2 * c16]     = h2f(vr[np + 2 * c16])     * h2f(p16[np + 2 * c16]);
        if (2 * c16 + 1 < ntail) tail[rl][2 * c16 + 1] = h2f(vr[np + 2 * c16 + 1]) * h2f(p16[np + 2 * c16 + 1]);
    }
    wave_lds_fence();
    if (c16 == 0) {
        double sacc = (double) res;
        for (int t = 0; t < ntail; t++) sacc += (double) tail[rl][t];
        att[h * HD + d0] = (float) sacc;
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
    int nsplit = 2 * cus / nkv; if (nsplit < 1) nsplit = 1; if (nsplit > 128) nsplit = 128;      // two 256-thread workgroups per CU
#define SC(HD_, MODE_, R2_) hipLaunchKernelGGL((k_attn_long_scores<HD_, MODE_, R2_>), dim3(nsplit, nkv), dim3(256), 0, st, qkv, pos_dev, rope_cs, nh, nkv, scale, k_cache, v_cache, (int) ML, S, g_long_ts)
#define SC2(HD_, R2_) do { if (mode == 0) SC(HD_, 0, R2_); else SC(HD_, 2, R2_); } while (0)
#define SC3(HD_) do { if (r2 == 1) SC2(HD_, 1); else if (r2 == 2) SC2(HD_, 2); else if (r2 == 4) SC2(HD_, 4); else SC2(HD_, 8); } while (0)
    if (hd == 128) SC3(128); else SC3(64);
    LAUNCH_CHECK();
    static const bool no_fuse = getenv("CLLM_ATTN_LONG_3") && atoi(getenv("CLLM_ATTN_LONG_3")) == 1;      // (tools: the three-launch form at every length)
    if (ML <= AL_FUSED_MAX_ML && ML % 32 == 0 && !no_fuse) {                       // soft_max inside the V.P launch
        const size_t lds2 = (size_t) ML * 4 + (size_t)(ML / 8) * 4 + (size_t) ML * 2 + 4 * AL_NS * 4096;
#define PV2(HD_) do { \
            static uint64_t attr2 = 0; \
            if (dev_flag_unset(attr2)) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_softmax_pv<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr2); } \
            hipLaunchKernelGGL((k_attn_long_softmax_pv<HD_>), dim3(HD_ / 16, nkv, r2), dim3(1024), lds2, st, pos_dev, nh, nkv, (const uint16_t *) v_cache, (int) ML, (const float *) S, att); } while (0)
        if (hd == 128) PV2(128); else PV2(64);
#undef PV2
        LAUNCH_CHECK();
        return CLLM_OK;
    }
    static uint64_t attr = 0;
    if (lds > 48 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_softmax, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr); }
    hipLaunchKernelGGL(k_attn_long_softmax, dim3(nh), dim3(1024), lds, st, pos_dev, (int) ML, (const float *) S, P16);
    LAUNCH_CHECK();
    const size_t lds_pv = (size_t) ML * 2 + 16;
    if (lds_pv > 150 * 1024) return CLLM_E_UNSUPPORTED;
#define PV(HD_) do { \
        static uint64_t attr_pv = 0; \
        if (lds_pv > 48 * 1024 && dev_flag_unset(attr_pv)) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_long_pv<HD_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr_pv); } \
        hipLaunchKernelGGL((k_attn_long_pv<HD_>), dim3(HD_ / 16, nkv, r2), dim3(256), lds_pv, st, pos_dev, nh, nkv, (const uint16_t *) v_cache, (int) ML, (const uint16_t *) P16, att); } while (0)
    if (hd == 128) PV(128); else PV(64);
    LAUNCH_CHECK();
#undef SC
#undef SC2
#undef SC3
#undef PV
    return CLLM_OK;
}
