// overlap_probe.hip -- can launch N + 1 start streaming its weights while launch N drains?  (tools/probes: measurement only, not part of the library)
// A synthetic decode mat-vec launch: 256 workgroups x 1024 threads, 120 KB LDS (one workgroup per CU), each workgroup (1) requests the head of its weight slice,
// (2) waits until the previous launch has signalled completion (device counter), (3) spends PRO_US of "activation prologue", (4) streams its slice, (5) signals.
// mode 0: every launch on ONE stream (the runtime's barrier between launches), no counter wait.
// mode 1: launches alternate between TWO streams; launch k waits on the device counter of launch k - 1 -- its workgroups can take the CUs that launch k - 1's
//         finished workgroups leave and issue their first requests while the stragglers of k - 1 still run.
// build: hipcc --offload-arch=gfx950 -O3 -o overlap_probe overlap_probe.hip ; run: ./overlap_probe [MB per launch = 66] [prologue us = 2]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) k_stage(const char * __restrict__ W, size_t wg_bytes, const unsigned * wait_ctr, unsigned target, unsigned * sig_ctr,
                                                unsigned * __restrict__ out, int pro_ticks, unsigned * err) {
    extern __shared__ char lds[];
    const int tid = threadIdx.x;
    const char * base = W + (size_t) blockIdx.x * wg_bytes;
    const size_t nvec = wg_bytes / 16;                                   // 16-byte vectors of this workgroup
    u32x4 pre[3];
#pragma unroll
    for (int p = 0; p < 3; p++) { const size_t i = (size_t) tid + (size_t) p * 1024; pre[p] = *(const u32x4 *)(base + (i < nvec ? i : 0) * 16); }
    if (wait_ctr) {
        if (tid == 0) {
            int spins = 0;
            while ((int)(__hip_atomic_load(wait_ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1 << 22)) { atomicOr(err, 1u); break; }
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    // the "prologue": pro_ticks x 10 ns of wall clock, then a barrier
    { const unsigned long long t0 = wall_clock64(); while (wall_clock64() - t0 < (unsigned long long) pro_ticks) __builtin_amdgcn_s_sleep(1); }
    ((unsigned *) lds)[tid] = pre[0].x;
    __syncthreads();
    unsigned acc = pre[0].x ^ pre[1].y ^ pre[2].z ^ ((unsigned *) lds)[(tid + 1) & 1023];
    for (size_t i = (size_t) tid + 3 * 1024; i < nvec; i += 8192) {
        u32x4 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const size_t j = i + (size_t) u * 1024; v[u] = __builtin_nontemporal_load((const u32x4 *)(base + (j < nvec ? j : 0) * 16)); }
#pragma unroll
        for (int u = 0; u < 8; u++) acc ^= v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 0x12345678u) out[tid] = acc;
    __syncthreads();
    if (tid == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); atomicAdd(sig_ctr, 1u); }
}

int main(int argc, char ** argv) {
    const size_t mb = argc > 1 ? atoi(argv[1]) : 66;
    const int pro_us = argc > 2 ? atoi(argv[2]) : 2;
    const int grid = 256, iters = 200, ncopy = (int)(1400 / mb) + 2;
    const size_t wg_bytes = (mb << 20) / grid / 16 * 16, bytes = wg_bytes * grid;
    std::vector<char *> w(ncopy);
    for (auto & p : w) { CHECK(hipMalloc((void **) &p, bytes)); CHECK(hipMemset(p, 0x5a, bytes)); }
    unsigned * ctr, * out, * err;
    CHECK(hipMalloc((void **) &ctr, 8)); CHECK(hipMalloc((void **) &out, 4096)); err = ctr + 1;
    CHECK(hipFuncSetAttribute((const void *) k_stage, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipStream_t s[2]; CHECK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    hipEvent_t e0, e1, ej; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1)); CHECK(hipEventCreate(&ej));
    const int nmodes = argc > 3 ? atoi(argv[3]) : 2;
    for (int mode = 0; mode < nmodes; mode++) for (int rep = 0; rep < 2; rep++) {
        CHECK(hipMemset(ctr, 0, 8)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, s[0]));
        if (mode == 1) { CHECK(hipStreamWaitEvent(s[1], e0, 0)); }
        for (int k = 0; k < iters; k++) {
            hipStream_t st = mode == 1 ? s[k & 1] : s[0];
            hipLaunchKernelGGL(k_stage, dim3(grid), dim3(1024), 120 * 1024, st, w[k % ncopy], wg_bytes, (mode == 1 && k > 0) ? ctr : (const unsigned *) nullptr, (unsigned)(k * grid), ctr, out,
                               pro_us * 100, err);
        }
        if (mode == 1) { CHECK(hipEventRecord(ej, s[1])); CHECK(hipStreamWaitEvent(s[0], ej, 0)); }
        CHECK(hipEventRecord(e1, s[0])); CHECK(hipEventSynchronize(e1));
        float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
        unsigned h[2]; CHECK(hipMemcpy(h, ctr, 8, hipMemcpyDeviceToHost));
        printf("mode %d (%s) rep %d: %.2f us per launch of %zu MB (%.0f GB/s), prologue %d us, counter %u (expected %u), timeouts %u\n", mode, mode ? "two streams + device counter" : "one stream", rep,
               ms * 1e3 / iters, mb, bytes / (ms * 1e-3 / iters) * 1e-9, pro_us, h[0], (unsigned)(iters * grid), h[1]);
    }
    return 0;
}
