/* oracle/glibc_math_check.c -- TEST INFRASTRUCTURE.  Compares chatllm.cpp_amd/csrc/glibc_math.h (compiled for the host) with the libm of
 * this image: prints the number of inputs whose float results differ.  Usage: glibc_math_check [n_random] */
#include "../chatllm.cpp_amd/csrc/glibc_math.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
static uint64_t rs = 88172645463325252ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 16); }
int main(int argc, char ** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    long bad_e = 0, bad_s = 0, bad_c = 0, tot = 0;
    for (long i = 0; i < n; i++) {
        float x = gm_asfloat(rnd());
        if (x != x) continue;
        tot++;
        const float e0 = expf(x), e1 = gm_expf(x), s0 = sinf(x), s1 = gm_sinf(x), c0 = cosf(x), c1 = gm_cosf(x);
        if (gm_asuint(e0) != gm_asuint(e1) && !(e0 != e0 && e1 != e1)) { if (bad_e++ < 5) fprintf(stderr, "expf(%a) = %a vs %a\n", x, e0, e1); }
        if (gm_asuint(s0) != gm_asuint(s1) && !(s0 != s0 && s1 != s1)) { if (bad_s++ < 5) fprintf(stderr, "sinf(%a) = %a vs %a\n", x, s0, s1); }
        if (gm_asuint(c0) != gm_asuint(c1) && !(c0 != c0 && c1 != c1)) { if (bad_c++ < 5) fprintf(stderr, "cosf(%a) = %a vs %a\n", x, c0, c1); }
    }
    /* the ranges the path actually uses, densely: soft_max / SiLU arguments in [-104, 90], RoPE angles in [0, 2^18] */
    for (long i = 0; i < n; i++) {
        const float xe = -104.0f + 194.0f * (float)(rnd() >> 8) / 16777216.0f;
        const float xa = ldexpf((float)(rnd() >> 8) / 16777216.0f, (int)(rnd() % 40) - 21);
        tot++;
        if (gm_asuint(expf(xe)) != gm_asuint(gm_expf(xe))) { if (bad_e++ < 5) fprintf(stderr, "expf(%a)\n", xe); }
        if (gm_asuint(sinf(xa)) != gm_asuint(gm_sinf(xa))) { if (bad_s++ < 5) fprintf(stderr, "sinf(%a) = %a vs %a\n", xa, sinf(xa), gm_sinf(xa)); }
        if (gm_asuint(cosf(xa)) != gm_asuint(gm_cosf(xa))) { if (bad_c++ < 5) fprintf(stderr, "cosf(%a) = %a vs %a\n", xa, cosf(xa), gm_cosf(xa)); }
    }
    printf("%ld inputs: expf %ld, sinf %ld, cosf %ld mismatches\n", tot, bad_e, bad_s, bad_c);
    return (bad_e || bad_s || bad_c) ? 1 : 0;
}
