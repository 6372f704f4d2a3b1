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
    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.f / maxv;
        q0 = min(127, nearest_int_dev(iscale * v.x));
        q1 = min(127, nearest_int_dev(iscale * v.y));
        q2 = min(127, nearest_int_dev(iscale * v.z));
        q3 = min(127, nearest_int_dev(iscale * v.w));
        d = 1 / iscale;
    }
    const int s = group8_sum_i(q0 + q1 + q2 + q3);
    *d_out = d;
    *s_out = s;
    return pack4(q0, q1, q2, q3);
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
