// mfma_f32_occupancy.hip -- v_mfma_f32_16x16x4_f32 issue interval per SIMD as the exact-order K.Q / V.P kernels use it: 16 accumulators (2 patches x 8 chains),
// DISTINCT operand registers per chain, 1 / 2 / 3 waves per SIMD, with and without the 24 fp16 -> f32 conversions per 16 MFMAs.
//   hipcc --offload-arch=gfx950 -O2 -Wno-unused-value -o tools/micro/bin/mfma_f32_occupancy tools/micro/mfma_f32_occupancy.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef uint32_t u4 __attribute__((ext_vector_type(4)));
#define ITERS 4096

template <int MODE>   // 0: operands fixed in registers; 1: + 24 conversions per step from packed halves that change every step; 2: conversions only (no MFMA)
__global__ void __launch_bounds__(256, 3) k_chain(const float * in, float * out) {
    f4 D[2][8];
    for (int j = 0; j < 2; j++) for (int L = 0; L < 8; L++) D[j][L] = (f4){0, 0, 0, 0};
    float xf[8], w0[8], w1[8];
    for (int i = 0; i < 8; i++) { xf[i] = in[threadIdx.x + i]; w0[i] = in[threadIdx.x + 8 + i]; w1[i] = in[threadIdx.x + 16 + i]; }
    u4 px = *(const u4 *)(in + threadIdx.x * 4), p0 = *(const u4 *)(in + 1024 + threadIdx.x * 4), p1 = *(const u4 *)(in + 2048 + threadIdx.x * 4);
    float sink = 0;
    for (int it = 0; it < ITERS; it++) {
        if (MODE >= 1) {
            px.x ^= (uint32_t)(it & 1); p0.y ^= (uint32_t)(it & 1); p1.z ^= (uint32_t)(it & 1);
            const h8v hx = __builtin_bit_cast(h8v, px), h0 = __builtin_bit_cast(h8v, p0), h1 = __builtin_bit_cast(h8v, p1);
#pragma unroll
            for (int i = 0; i < 8; i++) { xf[i] = (float) hx[i]; w0[i] = (float) h0[i]; w1[i] = (float) h1[i]; }
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < 8; i++) sink += xf[i] + w0[i] + w1[i];
        } else {
#pragma unroll
            for (int L = 0; L < 8; L++) D[0][L] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[L], w0[L], D[0][L], 0, 0, 0);
#pragma unroll
            for (int L = 0; L < 8; L++) D[1][L] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[L], w1[L], D[1][L], 0, 0, 0);
        }
    }
    float s = sink;
    for (int j = 0; j < 2; j++) for (int L = 0; L < 8; L++) s += D[j][L][0] + D[j][L][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> static void run(const char * name, int waves, const float * in, float * out, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * waves;
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_chain<MODE>, dim3(blocks), dim3(256), 0, 0, in, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per = ms * 1e-3 * ghz * 1e9 / ((double) ITERS * 16 * waves);
    printf("%-34s %d wave(s) per SIMD: %8.1f us   %6.1f cycles per MFMA slot and SIMD at %.2f GHz\n", name, waves, ms * 1e3, per, ghz);
}

int main() {
    float * in, * out; hipMalloc(&in, 1 << 20); hipMalloc(&out, 256 * 3 * 256 * 4);
    float * h = (float *) malloc(1 << 20);
    for (int pass = 0; pass < 2; pass++) {
    // pass 0: smooth operand values (few bits toggle); pass 1: random fp16 pairs / random floats in [-2, 2) -- the chip clocks to its power budget, the data decide the power
    uint32_t rs = 12345u;
    for (int i = 0; i < (1 << 18); i++) {
        if (pass == 0) h[i] = 0.5f + (i % 97) * 0.01f;
        else { rs = rs * 1664525u + 1013904223u; const uint32_t lo = 0x3000u + ((rs >> 8) & 0x0fffu) + ((rs >> 7) & 0x8000u), hi = 0x3000u + ((rs >> 20) & 0x0fffu) + ((rs >> 3) & 0x8000u); const uint32_t w = lo | (hi << 16); memcpy(&h[i], &w, 4); }
    }
    printf("---- operand data: %s\n", pass ? "random fp16 pairs (as floats: random mantissas, exponents around 2^-31)" : "smooth");
    hipMemcpy(in, h, 1 << 20, hipMemcpyHostToDevice);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); const double ghz = p.clockRate * 1e-6;
    for (int w = 1; w <= 3; w++) run<0>("16 MFMAs, operands in registers", w, in, out, ghz);
    for (int w = 1; w <= 3; w++) run<1>("16 MFMAs + 24 v_cvt_f32_f16", w, in, out, ghz);
    for (int w = 1; w <= 3; w++) run<2>("24 v_cvt_f32_f16 + 24 adds only", w, in, out, ghz);
    }
    return 0;
}
