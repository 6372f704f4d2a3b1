// q4k.h -- the Q4_K x Q8_K mat-vec of one wave, in the accumulation ORDER of the reference's x86 AVX2 branch
// (ggml_vec_dot_q4_K_q8_K, ggml-cpu/arch/x86/quants.c:1742-1822), so that the result is bit-identical to libggml-cpu.so:
//   per super-block i (in row order) and AVX lane A = 0..7 (dword A of every 32-byte chunk of qs / q8):
//       sumi[A] = sum over the four 64-weight chunks c of  sc[2c] * (q4l . q8l)[A] + sc[2c+1] * (q4h . q8h)[A]      (exact int32)
//       acc[A]  = fma(y.d * x.d, (float) sumi[A], acc[A])                                                            (serial over i)
//       acc_m[k] = fma(-y.d * x.dmin, (float)(m[2k] S[2k] + m[2k+1] S[2k+1]), acc_m[k]),  S = sums of q8 over the 32-weight sub-blocks
//   result = hsum_float_8(acc) + ((acc_m[0] + acc_m[2]) + (acc_m[1] + acc_m[3]))
// Shared by the multi-column kernel (mmvq.hip) and the decode kernel (gemv_decode.hip): one definition, one order.
//
// Work split (unchanged: it is what makes the loads coalesce): a group of 8 lanes owns one 144-byte super-block per step; all 8 lanes
// fetch the 16-byte header {d, dmin, 12 packed 6-bit scales/mins} (one broadcast request), lane j fetches qs[16j..16j+15] = chunk
// c = j/2, bytes 16(j&1).. of it = the dwords of AVX lanes 4(j&1) + k, k = 0..3.  The integer partial sums are reduce-scattered over
// the 8 lanes (any order: exact), so that lane j ends up with sumi[A(j)], A(j) = 4(j&1) + (j&2) + (j>>2).  The serial fp32 chain cannot
// be spread over the wave: every lane group writes its {float(sumi), d} and {float(prod), dmin} records to a per-wave LDS buffer and,
// every two steps (16 super-blocks), lanes 0..11 walk the records in block order -- lanes 0..7 carry acc[], lanes 8..11 acc_m[].
#pragma once
#include "common.h"

#define DPP_ROW_SHL4 0x104
#define DPP_ROW_SHR4 0x114
#define DPP_ROW_ROR8 0x128

// value of lane (l ^ 4) of the same 8 lanes (no single DPP pattern does it: two masked row shifts)
__device__ __forceinline__ int lane_xor4_i(int v) {
    const int r = __builtin_amdgcn_update_dpp(0, v, DPP_ROW_SHL4, 0xf, 0x5, false);      // lanes 0-3, 8-11 of a row <- lane + 4
    return __builtin_amdgcn_update_dpp(r, v, DPP_ROW_SHR4, 0xf, 0xA, false);             // lanes 4-7, 12-15        <- lane - 4
}
// compiler-level ordering of the wave's own LDS records (the LDS executes one wave's DS operations in order)
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- chain records: per wave, 8 pairs of super-blocks x 12 slots x {x_i, d_i, x_i+1, d_i+1} floats (+ 64 B so that the idle lanes
//      12..15 of a row read inside the buffer) ----
#define Q4K_PAIR_BYTES  192
#define Q4K_CHAIN_BYTES (8 * Q4K_PAIR_BYTES + 64)

struct q4k_sel {
    int sh16, sh8, a_off, s_idx;   // scale pair / min selectors, activation byte offset, index of this lane's sub-block sum
    int w_off, wm_off;             // record offsets of this lane inside the wave's chain buffer (step parity adds 4 pairs)
    bool hi, mhi, b2, b4, wm;
};

// lane-constant selectors (get_scale_min_k4, ggml-quants.c:703-711, after the utmp shuffle of arch/x86/quants.c:1773-1778)
__device__ __forceinline__ q4k_sel q4k_lane_sel(int lane) {
    q4k_sel L;
    const int j = lane & 7, g = lane >> 3, p = j >> 1;
    L.sh16 = (j & 2) * 8;                       // chunk c = j/2: scales (2c, 2c+1) = 16-bit field (c&1) of u0 / u1
    L.hi   = j >= 4;
    L.a_off = 64 * (j >> 1) + 16 * (j & 1);     // activation bytes of the low-nibble half; +32: the high-nibble half
    // mins: lanes (2p, 2p+1) own mins (2k, 2k+1) with k = [0, 2, 1, 3][p], so that acc_m's final adds are neighbour exchanges
    const int k = (p & 1) * 2 + (p >> 1), mi = 2 * k + (j & 1);
    L.s_idx = mi; L.mhi = mi >= 4; L.sh8 = (mi & 3) * 8;
    L.b2 = (j & 2) != 0; L.b4 = (j & 4) != 0; L.wm = (j & 1) == 0;
    L.w_off  = (g >> 1) * Q4K_PAIR_BYTES + j * 16 + (g & 1) * 8;
    L.wm_off = (g >> 1) * Q4K_PAIR_BYTES + (8 + p) * 16 + (g & 1) * 8;
    return L;
}

// one super-block of one activation column: h = block header, q = this lane's 16 quant bytes, ar = quantized activation row (act layout,
// common.h) in LDS, off_d / off_s = its scale / sub-block-sum planes, bb = super-block index (in range), ok = the block is real (masked
// blocks write zero records: fma(0, 0, acc) == acc), rec = the wave's chain buffer + 4 pairs for odd steps
__device__ __forceinline__ void q4k_emit(const u32x4 h, const u32x4 q, const char * ar, int off_d, int off_s, int bb, bool ok, const q4k_sel & L, char * rec) {
    const float d    = h2f((uint16_t)(h.x & 0xffff));
    const float dmin = h2f((uint16_t)(h.x >> 16));
    // 6-bit unpack: u0 = sc[0..3], u1 = sc[4..7], u2 = m[0..3], u3 = m[4..7]
    const uint32_t u0 = h.y & 0x3f3f3f3fu;
    const uint32_t u2 = h.z & 0x3f3f3f3fu;
    const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
    const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
    const uint32_t scp = (L.hi ? u1 : u0) >> L.sh16;
    const int sc_lo = (int)(scp & 0xff), sc_hi = (int)((scp >> 8) & 0xff);
    const int mj    = (int)(((L.mhi ? u3 : u2) >> L.sh8) & 0xff);
    const u32x4 al = *(const u32x4 *)(ar + bb * 256 + L.a_off);
    const u32x4 ah = *(const u32x4 *)(ar + bb * 256 + L.a_off + 32);
    const float yd = ((const float *)(ar + off_d))[bb];
    const int   ys = ((const int *)(ar + off_s))[bb * 8 + L.s_idx];
    // the four dwords of this lane = AVX lanes 4(j&1) + k of chunk j/2
    // (24-bit multiplies: scales < 64, |dot| <= 4 * 15 * 127 -- the 32-bit v_mul_lo_u32 / v_mad_u64_u32 the compiler picks otherwise run at a quarter rate)
    const uint32_t wq[8] = { q.x & 0x0f0f0f0fu, (q.x >> 4) & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu, (q.y >> 4) & 0x0f0f0f0fu,
                             q.z & 0x0f0f0f0fu, (q.z >> 4) & 0x0f0f0f0fu, q.w & 0x0f0f0f0fu, (q.w >> 4) & 0x0f0f0f0fu };
    const uint32_t wa[8] = { al.x, ah.x, al.y, ah.y, al.z, ah.z, al.w, ah.w };
    int dp[8];
    dot4z_x8(wq, wa, dp);
    const int t0 = __mul24(sc_lo, dp[0]) + __mul24(sc_hi, dp[1]);
    const int t1 = __mul24(sc_lo, dp[2]) + __mul24(sc_hi, dp[3]);
    const int t2 = __mul24(sc_lo, dp[4]) + __mul24(sc_hi, dp[5]);
    const int t3 = __mul24(sc_lo, dp[6]) + __mul24(sc_hi, dp[7]);
    // reduce-scatter over the four chunks (lanes j, j^2, j^4, j^6): lane j keeps dword k = (j&2) + (j>>2)
    int k0 = L.b2 ? t2 : t0, k1 = L.b2 ? t3 : t1;
    const int s0 = L.b2 ? t0 : t2, s1 = L.b2 ? t1 : t3;
    k0 += dpp_i<DPP_QUAD_XOR2>(s0); k1 += dpp_i<DPP_QUAD_XOR2>(s1);
    const int keep = L.b4 ? k1 : k0, send = L.b4 ? k0 : k1;
#ifdef Q4K_NOSCATTER     // (timing experiments only)
    const int sumi = (t0 + t1) + (t2 + t3) + (keep & 0) + (send & 0);
#else
    const int sumi = keep + lane_xor4_i(send);
#endif
    // mins: prod[k] = m[2k] S[2k] + m[2k+1] S[2k+1] on the lane pair
    int pm = __mul24(mj, ys);                   // |ys| <= 32 * 127
    pm += dpp_i<DPP_QUAD_XOR1>(pm);
    float x = (float) sumi, dd = yd * d, xm = (float) pm, dm = (-yd) * dmin;
    if (!ok) { dd = 0.0f; dm = 0.0f; }           // a masked block: fma(0, x, acc) == acc for the finite x it carries (two selects instead of four)
#ifdef Q4K_NOWRITE       // (timing experiments only)
    asm volatile("" :: "v"(x), "v"(dd), "v"(xm), "v"(dm), "v"(rec));
#else
    *(float2 *)(rec + L.w_off) = float2{x, dd};
    *(float2 *)(rec + L.wm_off) = float2{xm, dm};
#endif       // (both lanes of the pair hold the same record: no branch)
}

// ---- the same records from FOUR lanes per super-block (the decode mat-vec, gemv_decode.hip): lane j owns the whole 64-weight chunk j (32 quant
//      bytes = the low- and high-nibble dwords of all 8 AVX lanes), so the per-lane costs -- header unpack, scale selection, conversions, record
//      stores -- are paid once per 32 weight bytes instead of once per 16, the mins need no lane exchange (chunk j holds sub-blocks 2j, 2j+1 = pair
//      k = j), and the reduce-scatter runs inside a quad: lane j ends with sumi[A], A = 4 (j >> 1) + 2 (j & 1) + {0, 1}.  Same integers, same records.
struct q4k_sel4 { int sh16, a_off, s_off, w_off, wm_off; bool hi, b2, b1; };
__device__ __forceinline__ q4k_sel4 q4k_lane_sel4(int lane) {
    q4k_sel4 L;
    const int j = lane & 3, g = lane >> 2;
    L.hi = j >= 2; L.sh16 = (j & 1) * 16;
    L.a_off = 64 * j; L.s_off = 8 * j;
    L.b2 = (j & 2) != 0; L.b1 = (j & 1) != 0;
    const int slot = (j >> 1) | ((j & 1) << 1);                  // AVX lanes (A0, A0 + 1) sit in slots (slot, slot + 4) of [A0 A4 A2 A6 | A1 A5 A3 A7]
    const int mslot = ((j & 1) << 1) | (j >> 1);                 // acc_m[k = j] in [m0 m2 m1 m3]
    L.w_off  = (g >> 1) * Q4K_PAIR_BYTES + slot * 16 + (g & 1) * 8;
    L.wm_off = (g >> 1) * Q4K_PAIR_BYTES + (8 + mslot) * 16 + (g & 1) * 8;
    return L;
}
__device__ __forceinline__ void q4k_emit4(const u32x4 h, const u32x4 qa, const u32x4 qb, const char * ar, int off_d, int off_s, int bb, bool ok, const q4k_sel4 & L, char * rec) {
    const float d    = h2f((uint16_t)(h.x & 0xffff));
    const float dmin = h2f((uint16_t)(h.x >> 16));
    const uint32_t u0 = h.y & 0x3f3f3f3fu;
    const uint32_t u2 = h.z & 0x3f3f3f3fu;
    const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
    const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
    const uint32_t scp = (L.hi ? u1 : u0) >> L.sh16, mnp = (L.hi ? u3 : u2) >> L.sh16;
    const int sc_lo = (int)(scp & 0xff), sc_hi = (int)((scp >> 8) & 0xff), m_lo = (int)(mnp & 0xff), m_hi = (int)((mnp >> 8) & 0xff);
    const char * ab = ar + bb * 256 + L.a_off;
    const u32x4 al0 = *(const u32x4 *) ab, al1 = *(const u32x4 *)(ab + 16), ah0 = *(const u32x4 *)(ab + 32), ah1 = *(const u32x4 *)(ab + 48);
    const float yd = ((const float *)(ar + off_d))[bb];
    const u32x2 ys = *(const u32x2 *)(ar + off_s + bb * 32 + L.s_off);
    const uint32_t wl[8] = { qa.x & 0x0f0f0f0fu, qa.y & 0x0f0f0f0fu, qa.z & 0x0f0f0f0fu, qa.w & 0x0f0f0f0fu, qb.x & 0x0f0f0f0fu, qb.y & 0x0f0f0f0fu, qb.z & 0x0f0f0f0fu, qb.w & 0x0f0f0f0fu };
    const uint32_t wh[8] = { (qa.x >> 4) & 0x0f0f0f0fu, (qa.y >> 4) & 0x0f0f0f0fu, (qa.z >> 4) & 0x0f0f0f0fu, (qa.w >> 4) & 0x0f0f0f0fu,
                             (qb.x >> 4) & 0x0f0f0f0fu, (qb.y >> 4) & 0x0f0f0f0fu, (qb.z >> 4) & 0x0f0f0f0fu, (qb.w >> 4) & 0x0f0f0f0fu };
    const uint32_t xl[8] = { al0.x, al0.y, al0.z, al0.w, al1.x, al1.y, al1.z, al1.w };
    const uint32_t xh[8] = { ah0.x, ah0.y, ah0.z, ah0.w, ah1.x, ah1.y, ah1.z, ah1.w };
    int dl[8], dh[8], t[8];
    dot4z_x8(wl, xl, dl);
    dot4z_x8(wh, xh, dh);
#pragma unroll
    for (int A = 0; A < 8; A++) t[A] = __mul24(sc_lo, dl[A]) + __mul24(sc_hi, dh[A]);
    int k[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { const int keep = L.b2 ? t[4 + i] : t[i], send = L.b2 ? t[i] : t[4 + i]; k[i] = keep + dpp_i<DPP_QUAD_XOR2>(send); }
    int r[2];
#pragma unroll
    for (int i = 0; i < 2; i++) { const int keep = L.b1 ? k[2 + i] : k[i], send = L.b1 ? k[i] : k[2 + i]; r[i] = keep + dpp_i<DPP_QUAD_XOR1>(send); }
    const int pm = __mul24(m_lo, (int) ys.x) + __mul24(m_hi, (int) ys.y);        // prod[k = j] = m[2j] S[2j] + m[2j+1] S[2j+1]
    float dd = yd * d, dm = (-yd) * dmin;
    if (!ok) { dd = 0.0f; dm = 0.0f; }
    *(float2 *)(rec + L.w_off)      = float2{(float) r[0], dd};
    *(float2 *)(rec + L.w_off + 64) = float2{(float) r[1], dd};
    *(float2 *)(rec + L.wm_off)     = float2{(float) pm, dm};
}

// FREE-ORDER pricing form (gemv_free32.hip, CLLM_DECODE_FREE_ORDER=2 only -- never a product path: Q4_K's Q8_K activations put any other fp32 order 0.5 sigma away at depth,
// profiles/r06_prefill_mode_decomposition.txt): q4k_emit4's exact integer sums of this lane's 64-weight chunk, folded at once  dd (float) sum_A t[A] + dm (float) pm  -- the four
// lanes of a super-block and all its blocks meet in the wave's final sum.  What it measures: how much of the decode mat-vec's time is the reference's accumulation order.
__device__ __forceinline__ float q4k_block_free4(const u32x4 h, const u32x4 qa, const u32x4 qb, const char * ar, int off_d, int off_s, int bb, bool ok, const q4k_sel4 & L) {
    const float d    = h2f((uint16_t)(h.x & 0xffff));
    const float dmin = h2f((uint16_t)(h.x >> 16));
    const uint32_t u0 = h.y & 0x3f3f3f3fu;
    const uint32_t u2 = h.z & 0x3f3f3f3fu;
    const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
    const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
    const uint32_t scp = (L.hi ? u1 : u0) >> L.sh16, mnp = (L.hi ? u3 : u2) >> L.sh16;
    const int sc_lo = (int)(scp & 0xff), sc_hi = (int)((scp >> 8) & 0xff), m_lo = (int)(mnp & 0xff), m_hi = (int)((mnp >> 8) & 0xff);
    const char * ab = ar + bb * 256 + L.a_off;
    const u32x4 al0 = *(const u32x4 *) ab, al1 = *(const u32x4 *)(ab + 16), ah0 = *(const u32x4 *)(ab + 32), ah1 = *(const u32x4 *)(ab + 48);
    const float yd = ((const float *)(ar + off_d))[bb];
    const u32x2 ys = *(const u32x2 *)(ar + off_s + bb * 32 + L.s_off);
    const uint32_t wl[8] = { qa.x & 0x0f0f0f0fu, qa.y & 0x0f0f0f0fu, qa.z & 0x0f0f0f0fu, qa.w & 0x0f0f0f0fu, qb.x & 0x0f0f0f0fu, qb.y & 0x0f0f0f0fu, qb.z & 0x0f0f0f0fu, qb.w & 0x0f0f0f0fu };
    const uint32_t wh[8] = { (qa.x >> 4) & 0x0f0f0f0fu, (qa.y >> 4) & 0x0f0f0f0fu, (qa.z >> 4) & 0x0f0f0f0fu, (qa.w >> 4) & 0x0f0f0f0fu,
                             (qb.x >> 4) & 0x0f0f0f0fu, (qb.y >> 4) & 0x0f0f0f0fu, (qb.z >> 4) & 0x0f0f0f0fu, (qb.w >> 4) & 0x0f0f0f0fu };
    const uint32_t xl[8] = { al0.x, al0.y, al0.z, al0.w, al1.x, al1.y, al1.z, al1.w };
    const uint32_t xh[8] = { ah0.x, ah0.y, ah0.z, ah0.w, ah1.x, ah1.y, ah1.z, ah1.w };
    int sl = 0, sh = 0;
#pragma unroll
    for (int A = 0; A < 8; A++) { sl = dot4(wl[A], xl[A], sl); sh = dot4(wh[A], xh[A], sh); }
    const int tot = __mul24(sc_lo, sl) + __mul24(sc_hi, sh);
    const int pm = __mul24(m_lo, (int) ys.x) + __mul24(m_hi, (int) ys.y);
    const float v = __builtin_fmaf((-yd) * dmin, (float) pm, (yd * d) * (float) tot);
    return ok ? v : 0.0f;
}

// walk `npairs` (a multiple of 4) pairs of records in block order; l16 = lane & 15 (lanes 0..7: acc[], 8..11: acc_m[], 12..15: idle)
__device__ __forceinline__ void q4k_chain(const char * chain, int npairs, int l16, float & acc) {
    for (int q0 = 0; q0 < npairs; q0 += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *(const f32x4 *)(chain + (q0 + u) * Q4K_PAIR_BYTES + l16 * 16);
#pragma unroll
        for (int u = 0; u < 4; u++) { acc = __builtin_fmaf(v[u].y, v[u].x, acc); acc = __builtin_fmaf(v[u].w, v[u].z, acc); }
    }
}

// row result from the chain accumulators (valid in lane 0 of every 16 lanes): hsum_float_8 over lanes 0..7 -- the records sit in the
// slot order [A0 A4 A2 A6 | A1 A5 A3 A7], so the reference's three adds (x[i] + x[4+i]; [0]+[2], [1]+[3]; [0]+[1]) are neighbour
// exchanges -- plus, from lanes 8..11 (slots [m0 m2 m1 m3]), (m0 + m2) + (m1 + m3)
// TAIL 1: Q4_K (lanes 8..11 reduced as above); TAIL 2: one scalar chain in lane 8 (Q4_1's summs); TAIL 0: none
template <int TAIL>
__device__ __forceinline__ float chain_finish(float acc) {
    float h = acc;
    h = h + dpp_f<DPP_QUAD_XOR1>(h);
    h = h + dpp_f<DPP_QUAD_XOR2>(h);
    const float hs = h + dpp_f<DPP_HALF_MIRROR>(h);
    if (TAIL == 1) return hs + dpp_f<DPP_ROW_ROR8>(h);
    if (TAIL == 2) return hs + dpp_f<DPP_ROW_ROR8>(acc);
    return hs;
}
