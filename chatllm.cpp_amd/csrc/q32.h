// q32.h -- one 32-weight block of the Q4_0 / Q4_1 / Q8_0 mat-vec for one lane (shared by the plain kernel in mmvq.hip and the
// decode kernel in gemv_decode.hip: identical accumulation order on every path).
//   ggml_vec_dot_q4_0_q8_0 / ggml_vec_dot_q4_1_q8_1 / ggml_vec_dot_q8_0_q8_0 (ggml-cpu/quants.c:115-150, 152-186, 305-333):
//   exact int32 block dot product, fp32 scale + accumulate.  One lane owns one block: fp16 d [+ fp16 m] + 16 (nibbles) or 32 (Q8_0)
//   quant bytes, only 2-byte (Q4_1: 4-byte) aligned in memory (gfx950 serves the misaligned dwordx4 directly).
#pragma once
#include "common.h"

struct __attribute__((packed, aligned(2))) u16x8_u2 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) u32x4_u4 { uint32_t x, y, z, w; };

template <int FMT> struct q32_fmt {
    static constexpr bool IS_Q8 = FMT == CLLM_TYPE_Q8_0, IS_Q41 = FMT == CLLM_TYPE_Q4_1;
    static constexpr int  BS = IS_Q8 ? 34 : IS_Q41 ? 20 : 18;          // bytes per block
};

// One block's bytes -> registers, in two halves so that a kernel can issue the loads early and unpack when the data is needed.
// Q4_0 / Q8_0 blocks are 18 / 34 bytes: only 2-byte aligned, and a dwordx4 at a 2-byte aligned address is served far below the rate of
// an aligned one (measured through the whole decode: Q4_1, 20-byte blocks and 11 % MORE bytes, ran 22 % faster than Q4_0).  So the
// load is the 4-byte ALIGNED window that contains the block (5 / 9 dwords; `odd` = the block starts in the upper half of the first
// one), and q32_align() shifts the fields into place.
//   raw: a (+ b for Q8_0) = the first 4 (8) dwords of the window, t = its last dword [Q4_1: a = the 16 quant bytes, t = d | m << 16]
template <int FMT>
__device__ __forceinline__ void q32_load_raw(const char * bp, u32x4 & a, u32x4 & b, uint32_t & t, uint32_t & odd) {
    if constexpr (q32_fmt<FMT>::IS_Q41) {
        t = *(const uint32_t *) bp;
        const u32x4_u4 r0 = *(const u32x4_u4 *)(bp + 4);
        a = u32x4{r0.x, r0.y, r0.z, r0.w};
        odd = 0;
    } else {
        const uintptr_t p = (uintptr_t) bp;
        odd = (uint32_t)(p & 2);
        const char * wp = (const char *)(p & ~(uintptr_t) 3);
        const u32x4_u4 r0 = *(const u32x4_u4 *) wp;
        a = u32x4{r0.x, r0.y, r0.z, r0.w};
        if constexpr (q32_fmt<FMT>::IS_Q8) {
            const u32x4_u4 r1 = *(const u32x4_u4 *)(wp + 16);
            b = u32x4{r1.x, r1.y, r1.z, r1.w};
            t = *(const uint32_t *)(wp + 32);
        } else t = *(const uint32_t *)(wp + 16);
    }
}
// raw window -> h = fp16 d (low half) [Q4_1: fp16 m in the high half], q0 / q1 = quant bytes (q1: Q8_0 only)
template <int FMT>
__device__ __forceinline__ void q32_align(const u32x4 a, const u32x4 b, uint32_t t, uint32_t odd, uint32_t & h, u32x4 & q0, u32x4 & q1) {
    if constexpr (q32_fmt<FMT>::IS_Q41) { h = t; q0 = a; }
    else {
        const bool up = odd != 0;
        h = up ? a.x >> 16 : a.x & 0xffffu;
        auto sh = [](uint32_t hi, uint32_t lo) { return __builtin_amdgcn_alignbit(hi, lo, 16); };      // bytes 2..5 of the pair lo, hi
        if constexpr (q32_fmt<FMT>::IS_Q8) {
            q0 = up ? u32x4{a.y, a.z, a.w, b.x} : u32x4{sh(a.y, a.x), sh(a.z, a.y), sh(a.w, a.z), sh(b.x, a.w)};
            q1 = up ? u32x4{b.y, b.z, b.w, t}   : u32x4{sh(b.y, b.x), sh(b.z, b.y), sh(b.w, b.z), sh(t, b.w)};
        } else {
            q0 = up ? u32x4{a.y, a.z, a.w, t}   : u32x4{sh(a.y, a.x), sh(a.z, a.y), sh(a.w, a.z), sh(t, a.w)};
        }
    }
}
template <int FMT>
__device__ __forceinline__ void q32_load(const char * bp, uint32_t & h, u32x4 & q0, u32x4 & q1) {
    u32x4 a, b = {0, 0, 0, 0}; uint32_t t, odd;
    q32_load_raw<FMT>(bp, a, b, t, odd);
    q32_align<FMT>(a, b, t, odd, h, q0, q1);
}

// h / q0 / q1 as loaded above, ar = quantized activation row (Q8_0 / Q8_1 kind, common.h) in LDS, bb = block index (in range),
// ok = the block is real (not a masked dummy)
template <int FMT>
__device__ __forceinline__ void q32_step(uint32_t h, const u32x4 q0, const u32x4 q1, const char * ar, int off_d, int off_s, int bb, bool ok, float & acc) {
    const float d = h2f((uint16_t) h);
    const u32x4 a0 = *(const u32x4 *)(ar + bb * 32);          // elements 0..15  (nibble formats: <-> low nibbles)
    const u32x4 a1 = *(const u32x4 *)(ar + bb * 32 + 16);     // elements 16..31 (nibble formats: <-> high nibbles)
    const float yd = ((const float *)(ar + off_d))[bb];
    int s;
    float na;
    if constexpr (q32_fmt<FMT>::IS_Q8) {
        s = dot4(q0.x, a0.x, 0); s = dot4(q0.y, a0.y, s); s = dot4(q0.z, a0.z, s); s = dot4(q0.w, a0.w, s);
        s = dot4(q1.x, a1.x, s); s = dot4(q1.y, a1.y, s); s = dot4(q1.z, a1.z, s); s = dot4(q1.w, a1.w, s);
        na = __builtin_fmaf((float) s, d * yd, acc);
    } else {
        const uint32_t ql[4] = { q0.x & 0x0f0f0f0fu, q0.y & 0x0f0f0f0fu, q0.z & 0x0f0f0f0fu, q0.w & 0x0f0f0f0fu };
        const uint32_t qh[4] = { (q0.x >> 4) & 0x0f0f0f0fu, (q0.y >> 4) & 0x0f0f0f0fu, (q0.z >> 4) & 0x0f0f0f0fu, (q0.w >> 4) & 0x0f0f0f0fu };
        s = dot4(ql[0], a0.x, 0); s = dot4(ql[1], a0.y, s); s = dot4(ql[2], a0.z, s); s = dot4(ql[3], a0.w, s);
        s = dot4(qh[0], a1.x, s); s = dot4(qh[1], a1.y, s); s = dot4(qh[2], a1.z, s); s = dot4(qh[3], a1.w, s);
        if constexpr (q32_fmt<FMT>::IS_Q41) {                  // (d_w d_a) * sum nib*a  +  m_w * s_a
            const float ys = ((const float *)(ar + off_s))[bb];
            na = __builtin_fmaf(h2f((uint16_t)(h >> 16)), ys, __builtin_fmaf((float) s, d * yd, acc));
        } else {
            s -= 8 * ((const int *)(ar + off_s))[bb];          // sum (nib - 8) * a
            na = __builtin_fmaf((float) s, d * yd, acc);
        }
    }
    acc = ok ? na : acc;
}
