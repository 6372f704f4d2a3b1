// Does hipExtAnyOrderLaunch drop the AQL barrier bit on gfx950?  Chain of short kernels, in-order vs any-order.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
__global__ void k_work(const float* __restrict__ x, float* __restrict__ y, int n, int it) {
    // latency-bound: every workgroup just waits `it` ticks of the 100 MHz wall clock
    unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (unsigned long long)it) __builtin_amdgcn_s_sleep(1);
    if (threadIdx.x == 0) y[blockIdx.x] = x[blockIdx.x % n];
}
int main(int argc, char** argv) {
    int it = argc > 1 ? atoi(argv[1]) : 16;
    const int n = 1 << 22; float *x, *y;
    hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4);
    hipStream_t st; hipStreamCreate(&st);
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipStreamSynchronize(st);
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 400; ++i)
                hipExtLaunchKernelGGL(k_work, dim3(256), dim3(256), 0, st, nullptr, nullptr,
                                      mode ? hipExtAnyOrderLaunch : 0, x, y, n, it);
            hipStreamSynchronize(st);
            double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
            printf("mode %d rep %d: %.2f us per kernel\n", mode, rep, us / 400);
        }
    return 0;
}
