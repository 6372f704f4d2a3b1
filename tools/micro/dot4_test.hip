// micro-test: VOP3P v_dot4_i32_i8 with an inline 0 accumulator (one instruction) against the builtin (v_mov 0 + v_dot4c)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(const uint32_t * a, const uint32_t * b, int * o1, int * o2, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t x = a[i], y = b[i];
    o1[i] = __builtin_amdgcn_sdot4((int) x, (int) y, 0, false);
    int r;
    asm volatile("v_dot4_i32_i8 %0, %1, %2, 0" : "=v"(r) : "v"(x), "v"(y));
    o2[i] = r;
}
int main() {
    const int n = 1 << 16;
    uint32_t * a, * b; int * o1, * o2;
    hipMallocManaged(&a, n * 4); hipMallocManaged(&b, n * 4); hipMallocManaged(&o1, n * 4); hipMallocManaged(&o2, n * 4);
    uint32_t s = 12345;
    for (int i = 0; i < n; i++) { s = s * 1664525u + 1013904223u; a[i] = s & 0x0f0f0f0fu; s = s * 1664525u + 1013904223u; b[i] = s; }
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, a, b, o1, o2, n);
    hipDeviceSynchronize();
    int bad = 0;
    for (int i = 0; i < n; i++) {
        int ref = 0;
        for (int e = 0; e < 4; e++) ref += (int)(int8_t)(a[i] >> (8 * e)) * (int)(int8_t)(b[i] >> (8 * e));
        if (o1[i] != ref || o2[i] != ref) { if (bad < 5) printf("i=%d a=%08x b=%08x ref=%d builtin=%d asm=%d\n", i, a[i], b[i], ref, o1[i], o2[i]); bad++; }
    }
    printf("mismatches: %d of %d\n", bad, n);
    return bad != 0;
}
