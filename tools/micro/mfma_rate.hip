// mfma_rate.hip -- issue interval (cycles per instruction and SIMD) of the matrix-core / VALU instructions the exact-order prefill kernels are built from,
// one wave per SIMD, 8 independent accumulators:   hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/mfma_rate tools/micro/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f32v __attribute__((ext_vector_type(32)));
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16v __attribute__((ext_vector_type(16)));
typedef int i32v __attribute__((ext_vector_type(32)));
typedef float f2 __attribute__((ext_vector_type(2)));

#define ITERS 2048
#define BODY(NAME, ACC_T, NACC, INIT, STEP) \
__global__ void __launch_bounds__(256) NAME(const float * in, float * out) { \
    ACC_T acc[NACC]; \
    for (int q = 0; q < NACC; q++) acc[q] = INIT; \
    const h4 a4 = { (_Float16) in[threadIdx.x], (_Float16) 1, (_Float16) 2, (_Float16) 3 }, b4 = { (_Float16) in[threadIdx.x + 1], (_Float16) 1, (_Float16) 1, (_Float16) 2 }; \
    const h8 a8 = { a4[0], a4[1], a4[2], a4[3], a4[0], a4[1], a4[2], a4[3] }, b8 = { b4[0], b4[1], b4[2], b4[3], b4[0], b4[1], b4[2], b4[3] }; \
    const int ai = (int) in[threadIdx.x + 2], bi = (int) in[threadIdx.x + 3]; const float af = in[threadIdx.x + 4], bf = in[threadIdx.x + 5]; \
    (void) a4; (void) b4; (void) a8; (void) b8; (void) ai; (void) bi; (void) af; (void) bf; \
    for (int it = 0; it < ITERS; it++) { _Pragma("unroll") for (int q = 0; q < NACC; q++) { STEP; } } \
    float s = 0; for (int q = 0; q < NACC; q++) s += (float) acc[q][0]; \
    out[blockIdx.x * 256 + threadIdx.x] = s; }

BODY(k_16x16x4_4b_f16, f16v, 8, (f16v){0}, acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f16(a4, b4, acc[q], 0, 0, 0))
BODY(k_32x32x4_2b_f16, f32v, 4, (f32v){0}, acc[q] = __builtin_amdgcn_mfma_f32_32x32x4f16(a4, b4, acc[q], 0, 0, 0))
BODY(k_4x4x4_16b_f16, f4, 8, (f4){0}, acc[q] = __builtin_amdgcn_mfma_f32_4x4x4f16(a4, b4, acc[q], 0, 0, 0))
BODY(k_16x16x4_4b_i8, i16v, 8, (i16v){0}, acc[q] = __builtin_amdgcn_mfma_i32_16x16x4i8(ai, bi, acc[q], 0, 0, 0))
BODY(k_32x32x4_2b_i8, i32v, 4, (i32v){0}, acc[q] = __builtin_amdgcn_mfma_i32_32x32x4i8(ai, bi, acc[q], 0, 0, 0))
BODY(k_4x4x4_16b_i8, i4, 8, (i4){0}, acc[q] = __builtin_amdgcn_mfma_i32_4x4x4i8(ai, bi, acc[q], 0, 0, 0))
BODY(k_16x16x16_f16, f4, 8, (f4){0}, acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[q], 0, 0, 0))
BODY(k_16x16x32_f16, f4, 8, (f4){0}, acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[q], 0, 0, 0))
BODY(k_16x16x4_f32, f4, 8, (f4){0}, acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf, acc[q], 0, 0, 0))
BODY(k_fma_f32, f2, 8, (f2){0}, acc[q][0] = __builtin_fmaf(af, bf, acc[q][0]))
BODY(k_pk_fma_f32, f2, 8, (f2){0}, asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"((f2){af, bf}), "v"((f2){bf, af})))
BODY(k_dot4_i8, i4, 8, (i4){0}, acc[q][0] = __builtin_amdgcn_sdot4(ai, bi, acc[q][0], false))

template <typename K> static void run(const char * name, K kern, int per_iter, const float * in, float * out, double ghz) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, in, out);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-22s %8.1f us for %d x %d per wave, one wave per SIMD: %6.1f cycles per instruction at %.2f GHz\n", name, ms * 1e3, ITERS, per_iter, ms * 1e-3 * ghz * 1e9 / ((double) ITERS * per_iter), ghz);
}

int main() {
    float * in, * out; hipMalloc(&in, 4096); hipMemset(in, 0, 4096); hipMalloc(&out, 256 * 256 * 4);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate * 1e-6;
    printf("device clockRate %.2f GHz (the effective clock under load is lower)\n", ghz);
    run("16x16x4_4b_f16", k_16x16x4_4b_f16, 8, in, out, ghz);
    run("32x32x4_2b_f16", k_32x32x4_2b_f16, 4, in, out, ghz);
    run("4x4x4_16b_f16", k_4x4x4_16b_f16, 8, in, out, ghz);
    run("16x16x4_4b_i8", k_16x16x4_4b_i8, 8, in, out, ghz);
    run("32x32x4_2b_i8", k_32x32x4_2b_i8, 4, in, out, ghz);
    run("4x4x4_16b_i8", k_4x4x4_16b_i8, 8, in, out, ghz);
    run("16x16x16_f16 (legacy)", k_16x16x16_f16, 8, in, out, ghz);
    run("16x16x32_f16 (gfx950)", k_16x16x32_f16, 8, in, out, ghz);
    run("16x16x4_f32", k_16x16x4_f32, 8, in, out, ghz);
    run("v_fma_f32", k_fma_f32, 8, in, out, ghz);
    run("v_pk_fma_f32", k_pk_fma_f32, 8, in, out, ghz);
    run("v_dot4_i32_i8", k_dot4_i8, 8, in, out, ghz);
    return 0;
}
