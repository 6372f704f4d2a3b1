// q4k.h -- one step of the Q4_K x Q8_K mat-vec for one lane (shared by the plain kernel in mmvq.hip and the decode
// kernel in gemv_decode.hip, so that both accumulate every row in exactly the same order).
//   ggml_vec_dot_q4_K_q8_K (ggml-cpu/quants.c:550-623): integer block dot products (exact), fp32 scale + accumulate.
// Work split: a group of 8 lanes owns one 144-byte super-block per step; all 8 lanes fetch the 16-byte header
// {d, dmin, 12 packed 6-bit scales/mins} (one broadcast request), lane j fetches qs[16j..16j+15] (two 64-weight halves:
// low nibbles belong to sub-block 2*(j/2), high nibbles to 2*(j/2)+1) and owns min #j.
#pragma once
#include "common.h"

struct q4k_sel { int sh16, sh8, a_off; bool hi; int j; };

// lane-constant scale selectors (get_scale_min_k4, ggml-quants.c:703-711, after the utmp shuffle of quants.c:577-582)
__device__ __forceinline__ q4k_sel q4k_lane_sel(int lane) {
    q4k_sel L;
    const int j = lane & 7;
    L.j = j;
    L.sh16 = (j & 2) * 8;          // pair p=j/2: 16-bit field (p&1) of utmp[p>>1]
    L.sh8  = (j & 3) * 8;          // min j: byte (j&3) of utmp[2 + (j>>2)]
    L.hi   = j >= 4;
    L.a_off = 64 * (j >> 1) + 16 * (j & 1);     // activation bytes for the low-nibble half; +32 for the high half
    return L;
}

// h = block header, q = this lane's 16 quant bytes, ar = quantized activation row (act layout, common.h) in LDS,
// off_d / off_s = its scale / sub-block-sum planes, bb = super-block index (in range), ok = step is real (not a masked dummy)
// ONE accumulator per lane: acc += (d yd) * sum_sub sc*dot  -  (dmin yd) * m*bsum.  (Two accumulators -- the scale part and the mins part
// summed separately and subtracted at the end of the row, the structure of the reference's AVX2 loop -- cost a second wave reduction per
// row: with 4096-long rows, two steps each, that was a measurable share of the mat-vec.  The order of fp32 additions is a tolerance-level
// choice; every path shares this function, so they all stay bit-identical to each other.)
__device__ __forceinline__ void q4k_step(const u32x4 h, const u32x4 q, const char * ar, int off_d, int off_s, int bb, bool ok, const q4k_sel & L,
                                         float & acc) {
    const float d    = h2f((uint16_t)(h.x & 0xffff));
    const float dmin = h2f((uint16_t)(h.x >> 16));
    // 6-bit unpack: u0 = sc[0..3], u1 = sc[4..7], u2 = m[0..3], u3 = m[4..7]
    const uint32_t u0 = h.y & 0x3f3f3f3fu;
    const uint32_t u2 = h.z & 0x3f3f3f3fu;
    const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
    const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
    const uint32_t scp = (L.hi ? u1 : u0) >> L.sh16;
    const int sc_lo = (int)(scp & 0xff), sc_hi = (int)((scp >> 8) & 0xff);
    const int mj    = (int)(((L.hi ? u3 : u2) >> L.sh8) & 0xff);
    const uint32_t ql[4] = { q.x & 0x0f0f0f0fu, q.y & 0x0f0f0f0fu, q.z & 0x0f0f0f0fu, q.w & 0x0f0f0f0fu };
    const uint32_t qh[4] = { (q.x >> 4) & 0x0f0f0f0fu, (q.y >> 4) & 0x0f0f0f0fu, (q.z >> 4) & 0x0f0f0f0fu, (q.w >> 4) & 0x0f0f0f0fu };
    const u32x4 al = *(const u32x4 *)(ar + bb * 256 + L.a_off);
    const u32x4 ah = *(const u32x4 *)(ar + bb * 256 + L.a_off + 32);
    const float yd = ((const float *)(ar + off_d))[bb];
    const int   ys = ((const int *)(ar + off_s))[bb * 8 + L.j];
    int il = dot4(ql[0], al.x, 0); il = dot4(ql[1], al.y, il); il = dot4(ql[2], al.z, il); il = dot4(ql[3], al.w, il);
    int ih = dot4(qh[0], ah.x, 0); ih = dot4(qh[1], ah.y, ih); ih = dot4(qh[2], ah.z, ih); ih = dot4(qh[3], ah.w, ih);
    const int t = sc_lo * il + sc_hi * ih;
    const float nd = __builtin_fmaf(d * yd, (float) t, acc);
    const float na = __builtin_fmaf(-(dmin * yd), (float)(mj * ys), nd);
    acc = ok ? na : acc;
}
