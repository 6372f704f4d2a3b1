// valu_rate.hip -- the VALU ceiling of the exact-order prefill GEMM (mmx.hip), measured the way the kernel runs: several waves per SIMD, and fp32 fmas issued
// BESIDE matrix-core instructions.  Questions: (1) instructions per cycle and SIMD of v_fma_f32 / v_pk_fma_f32 at 1, 2, 4, 8 waves per SIMD (one wave alone is
// issue-limited); (2) what a block of N fmas costs next to one v_mfma_f32_16x16x4_4b_f16 (do they overlap? do packed fmas?).
//   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
#define ITERS 4096

// MODE 0: 16 v_fma_f32 per iteration; 1: 8 v_pk_fma_f32 (the same 16 x 64 fmas); NMFMA matrix instructions per iteration in front of them
template <int MODE, int NMFMA>
__global__ void __launch_bounds__(256) k_rate(const float * in, float * out) {
    float a[16]; f2 p[8];
    const float x = in[threadIdx.x & 63], y = in[(threadIdx.x & 63) + 1];
    const f2 xx = {x, y}, yy = {y, x};
#pragma unroll
    for (int q = 0; q < 16; q++) a[q] = (float) q;
#pragma unroll
    for (int q = 0; q < 8; q++) p[q] = f2{(float) q, (float) q};
    const h4 a4 = {(_Float16) x, (_Float16) 1, (_Float16) 2, (_Float16) 3}, b4 = {(_Float16) y, (_Float16) 1, (_Float16) 1, (_Float16) 2};
    f16v D[2] = {(f16v){0}, (f16v){0}};
    for (int it = 0; it < ITERS; it++) {
#pragma unroll
        for (int m = 0; m < NMFMA; m++) D[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f16(a4, b4, D[m & 1], 0, 0, 0);
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 16; q++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[q]) : "v"(x), "v"(y));
        } else {
#pragma unroll
            for (int q = 0; q < 8; q++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[q]) : "v"(xx), "v"(yy));
        }
    }
    float s = D[0][0] + D[1][0];
#pragma unroll
    for (int q = 0; q < 16; q++) s += a[q];
#pragma unroll
    for (int q = 0; q < 8; q++) s += p[q][0] + p[q][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

typedef float f32v __attribute__((ext_vector_type(32)));
// the exact GEMM's real ratio: 32 x 64 fmas per two 16x16x4_4b MFMAs -- or per ONE 32x32x4_2b MFMA (the same 8192 MACs in one instruction: does the VALU port lose less?)
template <int BIG>
__global__ void __launch_bounds__(256) k_mix32(const float * in, float * out) {
    float a[32];
    const float x = in[threadIdx.x & 63], y = in[(threadIdx.x & 63) + 1];
#pragma unroll
    for (int q = 0; q < 32; q++) a[q] = (float) q;
    const h4 a4 = {(_Float16) x, (_Float16) 1, (_Float16) 2, (_Float16) 3}, b4 = {(_Float16) y, (_Float16) 1, (_Float16) 1, (_Float16) 2};
    f16v D[2] = {(f16v){0}, (f16v){0}}; f32v E = (f32v){0};
    for (int it = 0; it < ITERS; it++) {
        if (BIG) E = __builtin_amdgcn_mfma_f32_32x32x4f16(a4, b4, E, 0, 0, 0);
        else { D[0] = __builtin_amdgcn_mfma_f32_16x16x4f16(a4, b4, D[0], 0, 0, 0); D[1] = __builtin_amdgcn_mfma_f32_16x16x4f16(a4, b4, D[1], 0, 0, 0); }
#pragma unroll
        for (int q = 0; q < 32; q++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[q]) : "v"(x), "v"(y));
    }
    float s = D[0][0] + D[1][0] + E[0];
#pragma unroll
    for (int q = 0; q < 32; q++) s += a[q];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> static void run32(const char * name, K kern, int wps, const float * in, float * out, int cus, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, in, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, in, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9, per = (double) ITERS * wps;
    printf("%-34s %d wave(s)/SIMD, 32x64 fmas per iteration: %7.1f us, %6.1f cycles per iteration and SIMD  (%5.2f fma-lanes / cycle / SIMD)\n", name, wps, ms * 1e3, cyc / per, 32.0 * 64.0 * per / cyc);
}

template <typename K> static void run(const char * name, K kern, int wps, int nmfma, const float * in, float * out, int cus, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, in, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(cus * wps), dim3(256), 0, 0, in, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9;                       // cycles of the launch at the nominal clock
    const double per_simd_iters = (double) ITERS * wps;              // iterations executed per SIMD
    printf("%-14s %d wave(s)/SIMD, %d MFMA + 16x64 fmas per iteration: %7.1f us, %6.1f cycles per iteration and SIMD  (%5.2f fma-lanes / cycle / SIMD%s)\n", name, wps, nmfma, ms * 1e3,
           cyc / per_simd_iters, 16.0 * 64.0 * per_simd_iters / cyc, nmfma ? "" : "");
}

int main() {
    float * in, * out; hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6; const int cus = p.multiProcessorCount;
    printf("device clockRate %.2f GHz, %d CUs (cycles at the nominal clock; the effective clock under load is lower)\n", ghz, cus);
    for (int wps : {1, 2, 4, 8}) {
        run("v_fma_f32", k_rate<0, 0>, wps, 0, in, out, cus, ghz);
        run("v_pk_fma_f32", k_rate<1, 0>, wps, 0, in, out, cus, ghz);
    }
    for (int wps : {1, 2, 3, 4}) {
        run("v_fma_f32", k_rate<0, 1>, wps, 1, in, out, cus, ghz);
        run("v_pk_fma_f32", k_rate<1, 1>, wps, 1, in, out, cus, ghz);
        run("v_fma_f32", k_rate<0, 2>, wps, 2, in, out, cus, ghz);
        run("v_pk_fma_f32", k_rate<1, 2>, wps, 2, in, out, cus, ghz);
    }
    for (int wps : {1, 2, 3}) {
        run32("2 x v_mfma_f32_16x16x4_4b_f16 +", k_mix32<0>, wps, in, out, cus, ghz);
        run32("1 x v_mfma_f32_32x32x4_2b_f16 +", k_mix32<1>, wps, in, out, cus, ghz);
    }
    return 0;
}
