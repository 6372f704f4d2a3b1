// mfma_probe.hip -- operand / result layouts of the K = 4 multi-block MFMAs and of the f32 MFMA on gfx950, checked against the layout the
// exact-order prefill kernels (mmx.hip, attn_exact.hip) assume; and the f32 MFMA's accumulation order (a k-ordered fmaf chain).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/mfma_probe tools/micro/mfma_probe.hip && tools/micro/bin/mfma_probe
// Hypotheses (A operand: lane -> (row i, block b) holding k = 0..3; B operand: lane -> (col j, block b); D: lane, reg -> (block, i, j)):
//   16x16x4_4b f16 : A lane l: i = l & 15, b = l >> 4;  B lane l: j = l & 15, b = l >> 4;  D reg v (0..15): b = v >> 2, i = 4 (l >> 4) + (v & 3), j = l & 15
//   32x32x4_2b f16 : A lane l: i = l & 31, b = l >> 5;  B likewise;                        D reg v (0..31): b = v >> 4, i = (v & 3) + 8 ((v & 15) >> 2) + 4 (l >> 5), j = l & 31
//   16x16x4 f32    : A lane l: i = l & 15, k = l >> 4;  B lane l: j = l & 15, k = l >> 4;  D reg v (0..3): i = 4 (l >> 4) + v, j = l & 15;  chain k = 0, 1, 2, 3 after C
//   32x32x2 f32    : A lane l: i = l & 31, k = l >> 5;  B likewise;                        D reg v (0..15): i = (v & 3) + 8 (v >> 2) + 4 (l >> 5), j = l & 31
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));

__global__ void k_16x16x4_4b(const h4 * a, const h4 * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f16v acc;
    for (int v = 0; v < 16; v++) acc[v] = c[l * 16 + v];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f16(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 16; v++) d[l * 16 + v] = acc[v];
}
__global__ void k_32x32x4_2b(const h4 * a, const h4 * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f32v acc;
    for (int v = 0; v < 32; v++) acc[v] = c[l * 32 + v];
    acc = __builtin_amdgcn_mfma_f32_32x32x4f16(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 32; v++) d[l * 32 + v] = acc[v];
}
__global__ void k_16x16x4_f32(const float * a, const float * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f4v acc;
    for (int v = 0; v < 4; v++) acc[v] = c[l * 4 + v];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) d[l * 4 + v] = acc[v];
}
__global__ void k_32x32x2_f32(const float * a, const float * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f16v acc;
    for (int v = 0; v < 16; v++) acc[v] = c[l * 16 + v];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[l], b[l], acc, 0, 0, 0);
    for (int v = 0; v < 16; v++) d[l * 16 + v] = acc[v];
}
// two chained f32 MFMAs into the same accumulator: the order across instructions is program order (k = 0..3 then 4..7)
__global__ void k_16x16x4_f32_x2(const float * a, const float * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f4v acc;
    for (int v = 0; v < 4; v++) acc[v] = c[l * 4 + v];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[l], b[l], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[64 + l], b[64 + l], acc, 0, 0, 0);
    for (int v = 0; v < 4; v++) d[l * 4 + v] = acc[v];
}
// v_pk_fma_f32 against two fmaf
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void k_pk_fma(const float * a, const float * b, const float * c, float * d) {
    const int l = threadIdx.x;
    f2v x = {a[2 * l], a[2 * l + 1]}, y = {b[2 * l], b[2 * l + 1]}, z = {c[2 * l], c[2 * l + 1]}, r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
    d[2 * l] = r.x; d[2 * l + 1] = r.y;
}

static float frand(unsigned & s) { s = s * 1664525u + 1013904223u; return (float)((int)(s >> 8) - (1 << 23)) / (float)(1 << 20); }
template <typename T> static T * dev(const T * h, size_t n) { T * p; hipMalloc(&p, n * sizeof(T)); hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice); return p; }

int main() {
    int bad_total = 0;
    unsigned seed = 12345;
    {   // ---- 16x16x4_4b f16, integer-valued operands (exact) ----
        _Float16 a[64 * 4], b[64 * 4]; float c[64 * 16], d[64 * 16];
        for (int i = 0; i < 256; i++) { seed = seed * 1664525u + 1013904223u; a[i] = (_Float16)(int)((seed >> 20) % 31) - 15; seed = seed * 1664525u + 1013904223u; b[i] = (_Float16)(int)((seed >> 20) % 255) - 127; }
        for (int i = 0; i < 1024; i++) c[i] = (float)(i % 7);
        h4 * da = (h4 *) dev(a, 256); h4 * db = (h4 *) dev(b, 256); float * dc = dev(c, 1024), * dd = dev(d, 1024);
        hipLaunchKernelGGL(k_16x16x4_4b, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {
            const int blk = v >> 2, i = 4 * (l >> 4) + (v & 3), j = l & 15;
            float want = c[l * 16 + v];
            for (int k = 0; k < 4; k++) want += (float) a[(blk * 16 + i) * 4 + k] * (float) b[(blk * 16 + j) * 4 + k];
            if (want != d[l * 16 + v]) bad++;
        }
        printf("v_mfma_f32_16x16x4_4b_f16 layout hypothesis: %s (%d / 1024 differ)\n", bad ? "WRONG" : "ok", bad);
        bad_total += bad;
    }
    {   // ---- 32x32x4_2b f16 ----
        _Float16 a[64 * 4], b[64 * 4]; float c[64 * 32], d[64 * 32];
        for (int i = 0; i < 256; i++) { seed = seed * 1664525u + 1013904223u; a[i] = (_Float16)(int)((seed >> 20) % 31) - 15; seed = seed * 1664525u + 1013904223u; b[i] = (_Float16)(int)((seed >> 20) % 255) - 127; }
        for (int i = 0; i < 2048; i++) c[i] = (float)(i % 5);
        h4 * da = (h4 *) dev(a, 256); h4 * db = (h4 *) dev(b, 256); float * dc = dev(c, 2048), * dd = dev(d, 2048);
        hipLaunchKernelGGL(k_32x32x4_2b, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 32; v++) {
            const int blk = v >> 4, i = (v & 3) + 8 * ((v & 15) >> 2) + 4 * (l >> 5), j = l & 31;
            float want = c[l * 32 + v];
            for (int k = 0; k < 4; k++) want += (float) a[(blk * 32 + i) * 4 + k] * (float) b[(blk * 32 + j) * 4 + k];
            if (want != d[l * 32 + v]) bad++;
        }
        printf("v_mfma_f32_32x32x4_2b_f16 layout hypothesis: %s (%d / 2048 differ)\n", bad ? "WRONG" : "ok", bad);
        bad_total += bad;
    }
    {   // ---- 16x16x4 f32: layout + k-ordered fmaf chain ----
        float a[128], b[128], c[256], d[256];
        for (int i = 0; i < 128; i++) { a[i] = frand(seed); b[i] = frand(seed); }
        for (int i = 0; i < 256; i++) c[i] = frand(seed) * 3.0f;
        float * da = dev(a, 128), * db = dev(b, 128), * dc = dev(c, 256), * dd = dev(d, 256);
        hipLaunchKernelGGL(k_16x16x4_f32, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad = 0, bad_rev = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
            const int i = 4 * (l >> 4) + v, j = l & 15;
            float w = c[l * 4 + v], wr = c[l * 4 + v];
            for (int k = 0; k < 4; k++) w = fmaf(a[k * 16 + i], b[k * 16 + j], w);
            for (int k = 3; k >= 0; k--) wr = fmaf(a[k * 16 + i], b[k * 16 + j], wr);
            if (memcmp(&w, &d[l * 4 + v], 4)) bad++;
            if (memcmp(&wr, &d[l * 4 + v], 4)) bad_rev++;
        }
        printf("v_mfma_f32_16x16x4_f32: D = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))) bit for bit: %s (%d / 256 differ; reversed k order: %d differ)\n", bad ? "NO" : "yes", bad, bad_rev);
        bad_total += bad;
        hipLaunchKernelGGL(k_16x16x4_f32_x2, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 4; v++) {
            const int i = 4 * (l >> 4) + v, j = l & 15;
            float w = c[l * 4 + v];
            for (int k = 0; k < 4; k++) w = fmaf(a[k * 16 + i], b[k * 16 + j], w);
            for (int k = 0; k < 4; k++) w = fmaf(a[64 + k * 16 + i], b[64 + k * 16 + j], w);
            if (memcmp(&w, &d[l * 4 + v], 4)) bad++;
        }
        printf("two chained v_mfma_f32_16x16x4_f32: an 8-step fmaf chain in program order: %s (%d / 256 differ)\n", bad ? "NO" : "yes", bad);
        bad_total += bad;
    }
    {   // ---- 32x32x2 f32 ----
        float a[128], b[128], c[1024], d[1024];
        for (int i = 0; i < 128; i++) { a[i] = frand(seed); b[i] = frand(seed); }
        for (int i = 0; i < 1024; i++) c[i] = frand(seed) * 3.0f;
        float * da = dev(a, 128), * db = dev(b, 128), * dc = dev(c, 1024), * dd = dev(d, 1024);
        hipLaunchKernelGGL(k_32x32x2_f32, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {
            const int i = (v & 3) + 8 * (v >> 2) + 4 * (l >> 5), j = l & 31;
            float w = c[l * 16 + v];
            for (int k = 0; k < 2; k++) w = fmaf(a[k * 32 + i], b[k * 32 + j], w);
            if (memcmp(&w, &d[l * 16 + v], 4)) bad++;
        }
        printf("v_mfma_f32_32x32x2_f32 layout + k-ordered chain: %s (%d / 1024 differ)\n", bad ? "NO" : "yes", bad);
        bad_total += bad;
    }
    {   // ---- v_pk_fma_f32 == fmaf per component ----
        float a[128], b[128], c[128], d[128];
        for (int i = 0; i < 128; i++) { a[i] = frand(seed); b[i] = frand(seed) * 1e-3f; c[i] = frand(seed) * 1e2f; }
        float * da = dev(a, 128), * db = dev(b, 128), * dc = dev(c, 128), * dd = dev(d, 128);
        hipLaunchKernelGGL(k_pk_fma, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 128; i++) { const float w = fmaf(a[i], b[i], c[i]); if (memcmp(&w, &d[i], 4)) bad++; }
        printf("v_pk_fma_f32 == fmaf per component: %s (%d / 128 differ)\n", bad ? "NO" : "yes", bad);
        bad_total += bad;
    }
    {   // ---- is the fp16 K = 4 MFMA a k-ordered fmaf chain on NON-integer data?  (products of two fp16 are exact in fp32; the question is where the sum is rounded) ----
        _Float16 a[64 * 4], b[64 * 4]; float c[64 * 16], d[64 * 16];
        for (int i = 0; i < 256; i++) { a[i] = (_Float16)(frand(seed) * 3.0f); b[i] = (_Float16)(frand(seed) * 0.37f); }
        for (int i = 0; i < 1024; i++) c[i] = frand(seed) * 100.0f;
        h4 * da = (h4 *) dev(a, 256); h4 * db = (h4 *) dev(b, 256); float * dc = dev(c, 1024), * dd = dev(d, 1024);
        hipLaunchKernelGGL(k_16x16x4_4b, dim3(1), dim3(64), 0, 0, da, db, dc, dd);
        hipMemcpy(d, dd, sizeof(d), hipMemcpyDeviceToHost);
        int bad_seq = 0, bad_rev = 0, bad_once = 0, bad_pair = 0;
        for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {
            const int blk = v >> 2, i = 4 * (l >> 4) + (v & 3), j = l & 15;
            const _Float16 * pa = a + (blk * 16 + i) * 4, * pb = b + (blk * 16 + j) * 4;
            float seq = c[l * 16 + v], rev = c[l * 16 + v];
            for (int k = 0; k < 4; k++) seq = fmaf((float) pa[k], (float) pb[k], seq);
            for (int k = 3; k >= 0; k--) rev = fmaf((float) pa[k], (float) pb[k], rev);
            double once = (double) c[l * 16 + v];
            for (int k = 0; k < 4; k++) once += (double) pa[k] * (double) pb[k];
            const float pair = (float)((double) c[l * 16 + v] + ((double) pa[0] * pb[0] + (double) pa[1] * pb[1])) ;
            const float pair2 = (float)((double) pair + ((double) pa[2] * pb[2] + (double) pa[3] * pb[3]));
            if (seq != d[l * 16 + v]) bad_seq++;
            if (rev != d[l * 16 + v]) bad_rev++;
            if ((float) once != d[l * 16 + v]) bad_once++;
            if (pair2 != d[l * 16 + v]) bad_pair++;
        }
        printf("v_mfma_f32_16x16x4_4b_f16 on fractional data: differs from the k-ordered fmaf chain in %d / 1024, from the reversed chain in %d, from one rounding of the exact 4-term sum in %d, from two 2-term steps in %d\n",
               bad_seq, bad_rev, bad_once, bad_pair);
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("HIP error\n"); return 2; }
    return bad_total ? 1 : 0;
}
