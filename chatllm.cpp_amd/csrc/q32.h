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

// ---- the fp32 accumulation in the ORDER of the reference's x86 AVX2 branches (arch/x86/quants.c:543-577 Q4_0, 701-760 Q4_1,
// 1012-1040 Q8_0; tinyBLAS_Q0_AVX, llamafile/sgemm.cpp:1346-1790, keeps the same per-element chain): per 32-weight block i (row order)
// and AVX lane A (elements 4A..4A+3):  acc[A] = fma(d_w d_a, (float) isum[A], acc[A]);  result = hsum_float_8(acc) [+ summs, Q4_1:
// summs = fma(m_w, s_a, summs) over the blocks].  One lane owns one block and produces its eight integer sums; the serial chains run in
// lanes 0..7 (lane 8: summs) over per-wave LDS records, one step (64 blocks) at a time -- see q4k.h for the scheme.
//   records of a wave: 9 rows of 64 floats (row = slot of AVX lane A, row 8 = Q4_1's s_a; 272-byte pitch: conflict-free ds_read_b128),
//   then d[64] and (Q4_1) m_w[64]
#include "q4k.h"
#define Q32_ROW_BYTES   272
#define Q32_CHAIN_BYTES (9 * Q32_ROW_BYTES + 512)

// h / q0 / q1 as loaded above, ar = quantized activation row (Q8_0 / Q8_1 kind, common.h) in LDS, bb = block index (in range),
// ok = the block is real (masked blocks write zero records), rec = the wave's record buffer
template <int FMT>
__device__ __forceinline__ void q32_emit(uint32_t h, const u32x4 q0, const u32x4 q1, const char * ar, int off_d, int off_s, int bb, bool ok, int lane, char * rec) {
    const float d = h2f((uint16_t) h);
    const u32x4 a0 = *(const u32x4 *)(ar + bb * 32);          // elements 0..15  (nibble formats: <-> low nibbles)  = AVX lanes 0..3
    const u32x4 a1 = *(const u32x4 *)(ar + bb * 32 + 16);     // elements 16..31 (nibble formats: <-> high nibbles) = AVX lanes 4..7
    const float yd = ((const float *)(ar + off_d))[bb];
    int s[8];
    if constexpr (q32_fmt<FMT>::IS_Q8) {
        s[0] = dot4(q0.x, a0.x, 0); s[1] = dot4(q0.y, a0.y, 0); s[2] = dot4(q0.z, a0.z, 0); s[3] = dot4(q0.w, a0.w, 0);
        s[4] = dot4(q1.x, a1.x, 0); s[5] = dot4(q1.y, a1.y, 0); s[6] = dot4(q1.z, a1.z, 0); s[7] = dot4(q1.w, a1.w, 0);
    } else {
        constexpr bool SUB8 = !q32_fmt<FMT>::IS_Q41;          // Q4_0: (nib - 8) . a = nib . a + (-8, -8, -8, -8) . a
        auto c0 = [&](uint32_t a) { return SUB8 ? dot4(0xf8f8f8f8u, a, 0) : 0; };
        s[0] = dot4(q0.x & 0x0f0f0f0fu, a0.x, c0(a0.x)); s[1] = dot4(q0.y & 0x0f0f0f0fu, a0.y, c0(a0.y));
        s[2] = dot4(q0.z & 0x0f0f0f0fu, a0.z, c0(a0.z)); s[3] = dot4(q0.w & 0x0f0f0f0fu, a0.w, c0(a0.w));
        s[4] = dot4((q0.x >> 4) & 0x0f0f0f0fu, a1.x, c0(a1.x)); s[5] = dot4((q0.y >> 4) & 0x0f0f0f0fu, a1.y, c0(a1.y));
        s[6] = dot4((q0.z >> 4) & 0x0f0f0f0fu, a1.z, c0(a1.z)); s[7] = dot4((q0.w >> 4) & 0x0f0f0f0fu, a1.w, c0(a1.w));
    }
    float * X = (float *) rec + lane;
    // slot of AVX lane A = 4(A&1) + (A&2) + (A>>2): [A0 A4 A2 A6 | A1 A5 A3 A7] (chain_finish, q4k.h)
    constexpr int R = Q32_ROW_BYTES / 4;
    X[0 * R] = ok ? (float) s[0] : 0.0f; X[1 * R] = ok ? (float) s[4] : 0.0f; X[2 * R] = ok ? (float) s[2] : 0.0f; X[3 * R] = ok ? (float) s[6] : 0.0f;
    X[4 * R] = ok ? (float) s[1] : 0.0f; X[5 * R] = ok ? (float) s[5] : 0.0f; X[6 * R] = ok ? (float) s[3] : 0.0f; X[7 * R] = ok ? (float) s[7] : 0.0f;
    X[9 * R] = ok ? d * yd : 0.0f;
    if constexpr (q32_fmt<FMT>::IS_Q41) {
        const float ys = ((const float *)(ar + off_s))[bb];
        X[8 * R] = ok ? ys : 0.0f;
        X[9 * R + 64] = ok ? h2f((uint16_t)(h >> 16)) : 0.0f;
    }
}

// walk the 64 blocks of one step in order; l16 = lane & 15 (lanes 0..7: acc[], lane 8: Q4_1's summs, others idle)
template <int FMT>
__device__ __forceinline__ void q32_chain(const char * chain, int l16, float & acc) {
    const char * xr = chain + (l16 < 8 ? l16 : 8) * Q32_ROW_BYTES;
    const char * dr = chain + 9 * Q32_ROW_BYTES + ((q32_fmt<FMT>::IS_Q41 && l16 == 8) ? 256 : 0);
    for (int i0 = 0; i0 < 64; i0 += 16) {
        f32x4 xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { xv[u] = *(const f32x4 *)(xr + (i0 + 4 * u) * 4); dv[u] = *(const f32x4 *)(dr + (i0 + 4 * u) * 4); }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            acc = __builtin_fmaf(dv[u].x, xv[u].x, acc); acc = __builtin_fmaf(dv[u].y, xv[u].y, acc);
            acc = __builtin_fmaf(dv[u].z, xv[u].z, acc); acc = __builtin_fmaf(dv[u].w, xv[u].w, acc);
        }
    }
}

// ---- the FREE-ORDER tier (opt-in, CLLM_DECODE_FREE_ORDER=1; gemv_free32.hip): the same exact int32 block dot products, folded as  d_w d_a (float) sumi  with ONE integer sum per
// block (the eight AVX-lane sums added as integers) into a per-lane fp32 partial sum over the lane's blocks (64 apart), the lanes summed by the wave at the end of the row.  NOT
// the reference's fp32 order: a tolerance tier by construction (what round 1's kernels were); it exists to price the exact order.
template <int FMT>
__device__ __forceinline__ float q32_block_free(uint32_t h, const u32x4 q0, const u32x4 q1, const char * ar, int off_d, int off_s, int bb, bool ok) {
    const float d = h2f((uint16_t) h);
    const u32x4 a0 = *(const u32x4 *)(ar + bb * 32);
    const u32x4 a1 = *(const u32x4 *)(ar + bb * 32 + 16);
    const float yd = ((const float *)(ar + off_d))[bb];
    int s = 0;
    if constexpr (q32_fmt<FMT>::IS_Q8) {
        s = dot4(q0.x, a0.x, s); s = dot4(q0.y, a0.y, s); s = dot4(q0.z, a0.z, s); s = dot4(q0.w, a0.w, s);
        s = dot4(q1.x, a1.x, s); s = dot4(q1.y, a1.y, s); s = dot4(q1.z, a1.z, s); s = dot4(q1.w, a1.w, s);
    } else {
        s = dot4(q0.x & 0x0f0f0f0fu, a0.x, s); s = dot4(q0.y & 0x0f0f0f0fu, a0.y, s); s = dot4(q0.z & 0x0f0f0f0fu, a0.z, s); s = dot4(q0.w & 0x0f0f0f0fu, a0.w, s);
        s = dot4((q0.x >> 4) & 0x0f0f0f0fu, a1.x, s); s = dot4((q0.y >> 4) & 0x0f0f0f0fu, a1.y, s); s = dot4((q0.z >> 4) & 0x0f0f0f0fu, a1.z, s); s = dot4((q0.w >> 4) & 0x0f0f0f0fu, a1.w, s);
        if constexpr (!q32_fmt<FMT>::IS_Q41) {          // Q4_0: (nib - 8) . a = nib . a - 8 sum(a)
            s = dot4(0xf8f8f8f8u, a0.x, s); s = dot4(0xf8f8f8f8u, a0.y, s); s = dot4(0xf8f8f8f8u, a0.z, s); s = dot4(0xf8f8f8f8u, a0.w, s);
            s = dot4(0xf8f8f8f8u, a1.x, s); s = dot4(0xf8f8f8f8u, a1.y, s); s = dot4(0xf8f8f8f8u, a1.z, s); s = dot4(0xf8f8f8f8u, a1.w, s);
        }
    }
    float v = (d * yd) * (float) s;
    if constexpr (q32_fmt<FMT>::IS_Q41) v = __builtin_fmaf(h2f((uint16_t)(h >> 16)), ((const float *)(ar + off_s))[bb], v);      // + m_w s_a
    return ok ? v : 0.0f;
}
