// q32.h -- one 32-weight block of the Q4_0 / Q8_0 x Q8_0 mat-vec for one lane (shared by the plain kernel in mmvq.hip and the
// decode kernel in gemv_decode.hip: identical accumulation order on every path).
//   ggml_vec_dot_q4_0_q8_0 / ggml_vec_dot_q8_0_q8_0 (ggml-cpu/quants.c:115-150, 305-333): exact int32 block dot product,
//   fp32 scale + accumulate.  One lane owns one block: fp16 d + 16 (Q4_0: nibbles) or 32 (Q8_0) quant bytes, only 2-byte
//   aligned in memory (gfx950 serves the misaligned dwordx4 directly).
#pragma once
#include "common.h"

struct __attribute__((packed, aligned(2))) u16x8_u2 { uint32_t x, y, z, w; };

// d16 = the block's fp16 scale bits, q0 / q1 = its quant bytes (q1: Q8_0 only), ar = quantized activation row (Q8_0-kind act
// layout, common.h) in LDS, bb = block index (in range), ok = the block is real (not a masked dummy)
template <bool IS_Q8>
__device__ __forceinline__ void q32_step(uint32_t d16, const u32x4 q0, const u32x4 q1, const char * ar, int off_d, int off_s, int bb, bool ok, float & acc) {
    const float d = h2f((uint16_t) d16);
    const u32x4 a0 = *(const u32x4 *)(ar + bb * 32);          // elements 0..15  (Q4_0: <-> low nibbles)
    const u32x4 a1 = *(const u32x4 *)(ar + bb * 32 + 16);     // elements 16..31 (Q4_0: <-> high nibbles)
    const float yd = ((const float *)(ar + off_d))[bb];
    int s;
    if (IS_Q8) {
        s = dot4(q0.x, a0.x, 0); s = dot4(q0.y, a0.y, s); s = dot4(q0.z, a0.z, s); s = dot4(q0.w, a0.w, s);
        s = dot4(q1.x, a1.x, s); s = dot4(q1.y, a1.y, s); s = dot4(q1.z, a1.z, s); s = dot4(q1.w, a1.w, s);
    } else {
        const int ys = ((const int *)(ar + off_s))[bb];
        const uint32_t ql[4] = { q0.x & 0x0f0f0f0fu, q0.y & 0x0f0f0f0fu, q0.z & 0x0f0f0f0fu, q0.w & 0x0f0f0f0fu };
        const uint32_t qh[4] = { (q0.x >> 4) & 0x0f0f0f0fu, (q0.y >> 4) & 0x0f0f0f0fu, (q0.z >> 4) & 0x0f0f0f0fu, (q0.w >> 4) & 0x0f0f0f0fu };
        s = dot4(ql[0], a0.x, 0); s = dot4(ql[1], a0.y, s); s = dot4(ql[2], a0.z, s); s = dot4(ql[3], a0.w, s);
        s = dot4(qh[0], a1.x, s); s = dot4(qh[1], a1.y, s); s = dot4(qh[2], a1.z, s); s = dot4(qh[3], a1.w, s);
        s -= 8 * ys;                                           // sum (nib - 8) * y
    }
    const float na = __builtin_fmaf((float) s, d * yd, acc);
    acc = ok ? na : acc;
}
