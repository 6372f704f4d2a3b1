// quant_dev.h -- device-side activation quantizers on 4 consecutive values per lane (shared by quantize.hip and the
// fused decode kernels).  Bit-exact restatements of
//   quantize_row_q8_0 (x86 AVX2 branch)  ggml/src/ggml-cpu/arch/x86/quants.c:290-345
//   quantize_row_q8_K_ref                ggml/src/ggml-quants.c:2555-2592 (nearest_int :436-441)
#pragma once
#include "common.h"

__device__ __forceinline__ uint32_t pack4(int q0, int q1, int q2, int q3) {
    return (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
}

// 8 consecutive lanes hold one 32-block (4 values each).  Returns the packed int8 quads; *d (fp16-rounded scale as
// f32) and *s (sum of the block's int8) are valid on every lane of the 8-lane group.
// Q81 (quantize_row_q8_1, arch/x86/quants.c:388-480): the same quants; *s = the bits of fp16(d * sum) as f32, d NOT yet rounded.
template <bool Q81 = false>
__device__ __forceinline__ uint32_t quant4_q8_0(f32x4 v, float * d_out, int * s_out) {
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = group8_max(amax);
    const float d  = amax / 127.f;
    const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
    const int q0 = (int) rintf(v.x * id), q1 = (int) rintf(v.y * id), q2 = (int) rintf(v.z * id), q3 = (int) rintf(v.w * id);
    const int s = group8_sum_i(q0 + q1 + q2 + q3);
    *d_out = h2f(f2h(d));              // the CPU stores d as fp16 and reads it back for the dot product
    *s_out = Q81 ? __float_as_int(h2f(f2h(d * (float) s))) : s;
    return pack4(q0, q1, q2, q3);
}

__device__ __forceinline__ int nearest_int_dev(float fval) {
    const float val = fval + 12582912.f;
    return (int)(__float_as_uint(val) & 0x007fffff) - 0x00400000;
}

// a whole wave (64 lanes x 4 values) holds one 256-block; lane l owns elements 4l..4l+3.
// *d valid on all lanes; *s = sum over this lane's 32-sub-block (valid on all 8 lanes of the group).
__device__ __forceinline__ uint32_t quant4_q8_K(f32x4 v, int lane, float * d_out, int * s_out) {
    // max picks the FIRST element of largest magnitude (quants.c:2562-2566): wave max of |x|, then the first lane that
    // holds it (ballot + find-first-set, scalar) and the first such element inside that lane
    (void) lane;
    const float a0 = fabsf(v.x), a1 = fabsf(v.y), a2 = fabsf(v.z), a3 = fabsf(v.w);
    const float amax = wave_max(fmaxf(fmaxf(a0, a1), fmaxf(a2, a3)));
    const float mine = a0 == amax ? v.x : a1 == amax ? v.y : a2 == amax ? v.z : v.w;
    const unsigned long long has = __ballot(a0 == amax || a1 == amax || a2 == amax || a3 == amax);
    const float maxv = lane_f(mine, (int) __builtin_ctzll(has));              // has != 0: some lane holds the maximum
    // the quants: nearest_int()'s own two roundings (t = iscale * x, t + 1.5 * 2^23), whose LOW BYTE is the two's-complement quant -- |t| <= 127 (1 + 2^-23) < 127.5, so the
    // reference's MIN(127, .) cannot bind for a finite non-zero maximum -- packed with three v_perm, summed with one v_dot4 against (1, 1, 1, 1) (q16_byte4 below)
    uint32_t packed = 0u;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.f / maxv;
        const uint32_t b0 = __float_as_uint(iscale * v.x + 12582912.f), b1 = __float_as_uint(iscale * v.y + 12582912.f);
        const uint32_t b2 = __float_as_uint(iscale * v.z + 12582912.f), b3 = __float_as_uint(iscale * v.w + 12582912.f);
        packed = __builtin_amdgcn_perm(__builtin_amdgcn_perm(b3, b2, 0x0c0c0400u), __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u), 0x05040100u);
        d = 1 / iscale;
    }
    *d_out = d;
    *s_out = group8_sum_i(dot4(packed, 0x01010101u, 0));
    return packed;
}

// ---- quantize_row_q8_K with SIXTEEN values per lane: a 16-lane DPP row holds one 256-block (lane p: elements 16 p .. 16 p + 15), a wave four blocks.
// Same bytes as quant4_q8_K (= quantize_row_q8_K_ref, ggml-quants.c:2555-2592) at ~40 % of its instructions per element -- the per-block work (reductions,
// the two IEEE divisions, the stores) is paid once per 16 values of a lane instead of once per 4, every reduction stays inside a DPP row (no readlane / ballot),
// and the per-element work shrinks to two roundings: t = iscale * x, u = t + 1.5 * 2^23 -- nearest_int()'s own addition -- whose LOW BYTE is the two's-complement
// quant (|t| <= 127 (1 + 2^-23) < 127.5: the reference's MIN(127, .) can never bind for a finite, non-zero maximum).
//   * the maximum: "the first element of largest magnitude" matters only through its SIGN; max = the row's largest value unless -min is larger; only when both
//     +amax and -amax occur (mx == -mn) is the first occurrence looked for (wave-uniform branch, rare).
//   * sub-block sums: v_dot4 of the packed quants with (1, 1, 1, 1), the lane pair's two halves added through DPP.
// d_out: valid on every lane of the row; s_out: the 32-element sub-block sum, valid on both lanes of the pair (p, p ^ 1).
__device__ __forceinline__ float row16_max(float v) { v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); return fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v)); }
__device__ __forceinline__ float row16_min(float v) { v = fminf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fminf(v, dpp_f<DPP_QUAD_XOR2>(v)); v = fminf(v, dpp_f<DPP_HALF_MIRROR>(v)); return fminf(v, dpp_f<DPP_ROW_MIRROR>(v)); }
__device__ __forceinline__ int   row16_min_i(int v) { v = min(v, dpp_i<DPP_QUAD_XOR1>(v)); v = min(v, dpp_i<DPP_QUAD_XOR2>(v)); v = min(v, dpp_i<DPP_HALF_MIRROR>(v)); return min(v, dpp_i<DPP_ROW_MIRROR>(v)); }
__device__ __forceinline__ int   row16_or_i(int v)  { v |= dpp_i<DPP_QUAD_XOR1>(v); v |= dpp_i<DPP_QUAD_XOR2>(v); v |= dpp_i<DPP_HALF_MIRROR>(v); return v | dpp_i<DPP_ROW_MIRROR>(v); }
__device__ __forceinline__ uint32_t q16_byte4(float iscale, f32x4 x) {
    const uint32_t b0 = __float_as_uint(iscale * x.x + 12582912.f), b1 = __float_as_uint(iscale * x.y + 12582912.f);
    const uint32_t b2 = __float_as_uint(iscale * x.z + 12582912.f), b3 = __float_as_uint(iscale * x.w + 12582912.f);
    const uint32_t lo = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u), hi = __builtin_amdgcn_perm(b3, b2, 0x0c0c0400u);      // (b0, b1, 0, 0), (b2, b3, 0, 0)
    return __builtin_amdgcn_perm(hi, lo, 0x05040100u);
}
__device__ __forceinline__ u32x4 quant16_q8_K(const f32x4 (&v)[4], int p, float * d_out, int * s_out) {
    float mx = fmaxf(fmaxf(v[0].x, v[0].y), fmaxf(v[0].z, v[0].w)), mn = fminf(fminf(v[0].x, v[0].y), fminf(v[0].z, v[0].w));
#pragma unroll
    for (int i = 1; i < 4; i++) {
        mx = fmaxf(mx, fmaxf(fmaxf(v[i].x, v[i].y), fmaxf(v[i].z, v[i].w)));
        mn = fminf(mn, fminf(fminf(v[i].x, v[i].y), fminf(v[i].z, v[i].w)));
    }
    mx = row16_max(mx); mn = row16_min(mn);
    const float amax = fmaxf(mx, -mn);
    float maxv = mx >= -mn ? mx : mn;
    const bool tie = mx == -mn && amax != 0.0f;
    if (__ballot(tie) != 0ull) {                // +amax and -amax both occur in some row of the wave: the sign of the FIRST one (quants.c:2562-2566); every lane takes part (DPP)
        int first = 256; float val = 0.0f;
#pragma unroll
        for (int i = 3; i >= 0; i--) {
            if (fabsf(v[i].w) == amax) { first = 16 * p + 4 * i + 3; val = v[i].w; }
            if (fabsf(v[i].z) == amax) { first = 16 * p + 4 * i + 2; val = v[i].z; }
            if (fabsf(v[i].y) == amax) { first = 16 * p + 4 * i + 1; val = v[i].y; }
            if (fabsf(v[i].x) == amax) { first = 16 * p + 4 * i + 0; val = v[i].x; }
        }
        const int rowfirst = row16_min_i(first);
        const float fv = __int_as_float(row16_or_i(first == rowfirst ? __float_as_int(val) : 0));      // exactly one lane of the row owns index rowfirst
        if (tie) maxv = fv;
    }
    u32x4 q = {0u, 0u, 0u, 0u};
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.f / maxv;
        q.x = q16_byte4(iscale, v[0]); q.y = q16_byte4(iscale, v[1]); q.z = q16_byte4(iscale, v[2]); q.w = q16_byte4(iscale, v[3]);
        d = 1 / iscale;
    }
    int s = dot4(q.x, 0x01010101u, 0);
    s = dot4(q.y, 0x01010101u, s); s = dot4(q.z, 0x01010101u, s); s = dot4(q.w, 0x01010101u, s);
    *s_out = s + dpp_i<DPP_QUAD_XOR1>(s);
    *d_out = d;
    return q;
}
// the 16-lane row's block into an act row (Q8_K kind): 16 quant bytes per lane, the pair's sub-block sum, the block scale
__device__ __forceinline__ void quant16_store(char * act_row, int64_t K, int blk, int p, const f32x4 (&v)[4]) {
    float d; int s;
    const u32x4 q = quant16_q8_K(v, p, &d, &s);
    *(u32x4 *)(act_row + blk * 256 + 16 * p) = q;
    if ((p & 1) == 0) ((int32_t *)(act_row + act_off_s(K, 256)))[blk * 8 + (p >> 1)] = s;
    if (p == 0) ((float *)(act_row + act_off_d(K)))[blk] = d;
}

// store one lane's quad into an act row (layout in common.h); e0 = element index of the lane's first value
template <int KIND>    // 32: Q8_0 kind, 256: Q8_K kind
__device__ __forceinline__ void act_store(char * act_row, int64_t K, int64_t e0, int lane, uint32_t packed, float d, int s) {
    *(uint32_t *)(act_row + e0) = packed;
    if ((lane & 7) == 0) ((int32_t *)(act_row + act_off_s(K, KIND)))[e0 / 32] = s;
    if (KIND == 32) { if ((lane & 7) == 0) ((float *)(act_row + act_off_d(K)))[e0 / 32] = d; }
    else            { if ((lane & 63) == 0) ((float *)(act_row + act_off_d(K)))[e0 / 256] = d; }
}

template <int KIND, bool Q81 = false>
__device__ __forceinline__ void quant4_store(char * act_row, int64_t K, int64_t e0, int lane, f32x4 v) {
    float d; int s; uint32_t p;
    if (KIND == 32) p = quant4_q8_0<Q81>(v, &d, &s); else p = quant4_q8_K(v, lane, &d, &s);
    act_store<KIND>(act_row, K, e0, lane, p, d, s);
}
