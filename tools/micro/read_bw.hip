// read_bw.hip -- what does a pure streaming read reach on this MI355X?  (context for the GEMV roofline fraction)
// Each lane accumulates 16-byte loads; variants: workgroup size, workgroups per CU, loads in flight per lane (unroll), chunking.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/read_bw tools/micro/read_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int U>
__global__ void __launch_bounds__(1024) k_read(const uint4 * __restrict__ p, size_t n16, unsigned * out) {
    unsigned acc = 0;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; i < n16; i += stride) { const uint4 v = p[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;       // never true in practice; keeps the loads alive
}

// each WAVE streams its own contiguous chunk (the GEMV's access pattern: one weight row per wave)
template <int U>
__global__ void __launch_bounds__(1024) k_read_rows(const uint4 * __restrict__ p, size_t n16, size_t row16, unsigned * out) {
    unsigned acc = 0;
    const size_t nrows = n16 / row16;
    const size_t wave = ((size_t) blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((size_t) gridDim.x * blockDim.x) >> 6;
    const int lane = threadIdx.x & 63;
    for (size_t r = wave; r < nrows; r += nwaves) {
        const uint4 * row = p + r * row16;
        for (size_t i = lane; i < row16; i += 64 * U) {
            uint4 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = (i + u * 64 < row16) ? row[i + u * 64] : uint4{0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < U; u++) acc += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
        }
    }
    if (acc == 0x12345678u) out[0] = acc;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const size_t bytes = (size_t) 2 << 30, n16 = bytes / 16;
    uint4 * p; unsigned * out;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&out, 4)); CK(hipMemset(p, 1, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char * what) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int i = 0; i < 5; i++) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-52s %8.1f GB/s\n", what, bytes * 5 / (ms * 1e-3) / 1e9);
    };
    char name[128];
    for (int wg : {256, 1024}) for (int per_cu : {2, 4, 8}) {
        if (wg * per_cu > 2048) continue;
        const int grid = cus * per_cu;
        snprintf(name, sizeof name, "grid-stride  wg %4d x %d/CU  U=1", wg, per_cu); time([&] { hipLaunchKernelGGL(k_read<1>, dim3(grid), dim3(wg), 0, 0, p, n16, out); }, name);
        snprintf(name, sizeof name, "grid-stride  wg %4d x %d/CU  U=4", wg, per_cu); time([&] { hipLaunchKernelGGL(k_read<4>, dim3(grid), dim3(wg), 0, 0, p, n16, out); }, name);
        snprintf(name, sizeof name, "grid-stride  wg %4d x %d/CU  U=8", wg, per_cu); time([&] { hipLaunchKernelGGL(k_read<8>, dim3(grid), dim3(wg), 0, 0, p, n16, out); }, name);
    }
    for (size_t row_bytes : {2304, 8064}) for (int per_cu : {1, 2}) {
        const size_t row16 = row_bytes / 16; const int grid = cus * per_cu;
        snprintf(name, sizeof name, "row-per-wave %zu B rows wg 1024 x %d/CU U=1", row_bytes, per_cu); time([&] { hipLaunchKernelGGL(k_read_rows<1>, dim3(grid), dim3(1024), 0, 0, p, n16, row16, out); }, name);
        snprintf(name, sizeof name, "row-per-wave %zu B rows wg 1024 x %d/CU U=2", row_bytes, per_cu); time([&] { hipLaunchKernelGGL(k_read_rows<2>, dim3(grid), dim3(1024), 0, 0, p, n16, row16, out); }, name);
        snprintf(name, sizeof name, "row-per-wave %zu B rows wg 1024 x %d/CU U=4", row_bytes, per_cu); time([&] { hipLaunchKernelGGL(k_read_rows<4>, dim3(grid), dim3(1024), 0, 0, p, n16, row16, out); }, name);
    }
    return 0;
}
