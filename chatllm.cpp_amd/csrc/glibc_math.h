// glibc_math.h -- expf / sinf / cosf with the RESULTS of glibc 2.35's libm (the reference's CPU path calls them: the n mod 8 tail of
// ggml_vec_soft_max_f32 / ggml_vec_silu_f32, vec.cpp:396-431, 547-; the RoPE cache, ops.cpp:5589-5628), so that the device agrees with
// the host to the last bit where the device's own math library is only ~1 ulp close.  Restatement of the published algorithms of
// sysdeps/ieee754/flt-32/{e_expf.c, s_sinf.c, s_cosf.c, sincosf.h} (ARM optimized routines): everything is evaluated in double and
// rounded once to float, so only the operations' order matters, not the instruction selection.  Compiles for the host too:
// oracle/glibc_math_check.c compares it with the libm of this image over the float range (tests/test_oracle_vs_reference.py).
#pragma once
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define GM_FN __host__ __device__ static inline
#define GM_TAB static __device__ const
#define GM_TAB_HOST static const
#else
#define GM_FN static inline
#define GM_TAB static const
#endif

GM_FN uint32_t gm_asuint(float f)    { uint32_t u; memcpy(&u, &f, 4); return u; }
GM_FN float    gm_asfloat(uint32_t u){ float f; memcpy(&f, &u, 4); return f; }
GM_FN uint64_t gm_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
GM_FN double   gm_asdouble(uint64_t u){ double f; memcpy(&f, &u, 8); return f; }

// ---- expf (e_expf.c, exp2f_data.c: N = 32) -----------------------------------------------------------------------------------------
// T[i] = asuint64(2^(i/32)) - (i << 47); a table in constant memory on the device (a 32-way switch compiles to a branch tree that costs the
// one lane running a soft_max tail ~100 cycles per element)
#if defined(__HIPCC__)
__device__ __constant__ static const uint64_t gm_exp2f_T_dev[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull
};
#endif
static const uint64_t gm_exp2f_T_host[32] = {
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull
};
GM_FN uint64_t gm_exp2f_tab(int i) {
#if defined(__HIP_DEVICE_COMPILE__)
    return gm_exp2f_T_dev[i & 31];
#else
    return gm_exp2f_T_host[i & 31];
#endif
}
// in two steps so that a device caller can fetch the table entry T[ki & 31] its own way between them (from LDS: decode_fused.hip) -- the arithmetic and its order are
// those of the one-piece function
GM_FN uint64_t gm_expf_ki(float x) {
    // z = InvLn2N * xd feeds two additions only: gcc fuses both (kd = z + SHIFT, r = z - kd) and z itself is never rounded
    return gm_asuint64(__builtin_fma(0x1.71547652b82fep+0 * 32, (double) x, 0x1.8p+52));
}
GM_FN float gm_expf_fin(float x, uint64_t ki, uint64_t t /* T[ki & 31] */) {
    const double xd = (double) x;
    const uint32_t abstop = (gm_asuint(x) >> 20) & 0x7ff;
    if (abstop >= 0x42b) {                                  // |x| >= 88 or nan
        if (gm_asuint(x) == 0xff800000u) return 0.0f;
        if (abstop >= 0x7f8) return x + x;
        if (x > 0x1.62e42ep6f) return gm_asfloat(0x7f800000u);                  // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;                                    // underflow
        if (x < -0x1.9d1d9ep6f) return gm_asfloat(1u);                          // may-underflow: 0x1.4p-75f squared = the smallest subnormal
    }
    const double kd = gm_asdouble(ki) - 0x1.8p+52;
    const double r = __builtin_fma(0x1.71547652b82fep+0 * 32, xd, -kd);
    double z;
    t += ki << 47;
    const double s = gm_asdouble(t);
    // (the libm of this image runs its FMA build -- ifunc on x86-64-v3 hosts: the three multiply-adds are single roundings)
    z = __builtin_fma(0x1.c6af84b912394p-5 / 32 / 32 / 32, r, 0x1.ebfce50fac4f3p-3 / 32 / 32);
    const double r2 = r * r;
    double y = __builtin_fma(0x1.62e42ff0c52d6p-1 / 32, r, 1.0);
    y = __builtin_fma(z, r2, y);
    y = y * s;
    return (float) y;
}
GM_FN float gm_expf(float x) { const uint64_t ki = gm_expf_ki(x); return gm_expf_fin(x, ki, gm_exp2f_tab((int)(ki & 31))); }

// ---- sinf / cosf (s_sinf.c, s_cosf.c, sincosf.h, sincosf_data.c) -----------------------------------------------------------------------
GM_FN uint32_t gm_inv_pio4(int i) {        // the bits of 4/pi, 8 hex digits from every second digit on
    switch (i) {
        case 0: return 0xa2u; case 1: return 0xa2f9u; case 2: return 0xa2f983u; case 3: return 0xa2f9836eu; case 4: return 0xf9836e4eu; case 5: return 0x836e4e44u;
        case 6: return 0x6e4e4415u; case 7: return 0x4e441529u; case 8: return 0x441529fcu; case 9: return 0x1529fc27u; case 10: return 0x29fc2757u; case 11: return 0xfc2757d1u;
        case 12: return 0x2757d1f5u; case 13: return 0x57d1f534u; case 14: return 0xd1f534ddu; case 15: return 0xf534ddc0u; case 16: return 0x34ddc0dbu; case 17: return 0xddc0db62u;
        case 18: return 0xc0db6295u; case 19: return 0xdb629599u; case 20: return 0x6295993cu; case 21: return 0x95993c43u; case 22: return 0x993c4390u; default: return 0x3c439041u;
    }
}
// polynomial on [-pi/4, pi/4]; neg: the second table (c coefficients negated); odd n: cosine polynomial
GM_FN float gm_sincos_poly(double x, double x2, int neg, int n) {
    const double sg = neg ? -1.0 : 1.0;
    if ((n & 1) == 0) {
        const double s1c = -0x1.555545995a603p-3, s2c = 0x1.1107605230bc4p-7, s3c = -0x1.994eb3774cf24p-13;
        const double x3 = x * x2;
        const double s1 = s2c + x2 * s3c;
        const double x7 = x3 * x2;
        const double s = x + x3 * s1c;
        return (float)(s + x7 * s1);
    } else {
        const double c0 = sg * 0x1p0, c1c = sg * -0x1.ffffffd0c621cp-2, c2c = sg * 0x1.55553e1068f19p-5, c3c = sg * -0x1.6c087e89a359dp-10, c4c = sg * 0x1.99343027bf8c3p-16;
        const double x4 = x2 * x2;
        const double c2 = c3c + x2 * c4c;
        const double c1 = c0 + x2 * c1c;
        const double x6 = x4 * x2;
        const double c = c1 + x4 * c2c;
        return (float)(c + x6 * c2);
    }
}
GM_FN double gm_reduce_fast(double x, int * np) {
    const double r = x * 0x1.45F306DC9C883p+23;
    const int n = ((int32_t) r + 0x800000) >> 24;
    *np = n;
    return __builtin_fma(-(double) n, 0x1.921FB54442D18p0, x);      // the libm of this image runs its FMA build (ifunc): x - n * hpi is one rounding
}
GM_FN double gm_reduce_large(uint32_t xi, int * np) {
    const int a = (int)((xi >> 26) & 15);
    const int shift = (int)((xi >> 23) & 7);
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = (uint32_t)(xi * gm_inv_pio4(a));
    res1 = (uint64_t) xi * gm_inv_pio4(a + 4);
    res2 = (uint64_t) xi * gm_inv_pio4(a + 8);
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ull << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t) res0;
    *np = (int) n;
    return x * 0x1.921FB54442D18p-62;
}
GM_FN double gm_sign(int n) { return ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; }     // sign[4] = { 1, -1, -1, 1 }
// which: 0 = sinf, 1 = cosf
GM_FN float gm_sincosf(float y, int which) {
    double x = (double) y;
    const uint32_t at = (gm_asuint(y) >> 20) & 0x7ff;
    int n;
    if (at < 0x3f4) {                                       // |y| < pi/4   (abstop12(0x1.921FB6p-1f) = 0x3f4)
        if (at < 0x398) return which ? 1.0f : y;            // |y| < 2^-12
        return gm_sincos_poly(x, x * x, 0, which);
    } else if (at < 0x42f) {                                // |y| < 120    (abstop12(120.0f) = 0x42f)
        x = gm_reduce_fast(x, &n);
        const double s = gm_sign(n);
        return gm_sincos_poly(x * s, x * x, (n & 2) != 0, n ^ which);
    } else if (at < 0x7f8) {
        const uint32_t xi = gm_asuint(y);
        const int sign = (int)(xi >> 31);
        x = gm_reduce_large(xi, &n);
        const double s = gm_sign(n + sign);
        return gm_sincos_poly(x * s, x * x, ((n + sign) & 2) != 0, n ^ which);
    }
    return y - y;                                           // inf / nan -> nan
}
GM_FN float gm_sinf(float y) { return gm_sincosf(y, 0); }
GM_FN float gm_cosf(float y) { return gm_sincosf(y, 1); }
