// barrier_bench.hip -- micro-benchmark: cost of a device-wide barrier between persistent workgroups on MI355X
// (8 XCDs, private L2s), with and without a small all-to-all data exchange through agent-coherent loads/stores.
// Answers: is one persistent decode kernel with grid barriers cheaper than ~4.5 us per dependent kernel launch?
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/build/barrier_bench tools/micro/barrier_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// monotone counter barrier; bounded spin so that a bug cannot hang the GPU
// variant 0: release add / acquire polling + s_sleep     1: relaxed add, relaxed polling
// variant 2: 1 + one release fence before / one acquire fence after     3: two-level (per-XCD counter, then global), relaxed
// variant 4: flag array, no read-modify-write: workgroup b stores epoch it+1 into flags[b]; wave 0 of every workgroup polls all
//            flags (4 per lane, two 8-byte agent-coherent loads) until none is behind
__device__ __forceinline__ bool flag_barrier(unsigned * flags, unsigned epoch, unsigned * abort_flag) {
    __syncthreads();
    __shared__ int okflag;
    if (threadIdx.x < 64) {
        if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned n = gridDim.x;                     // <= 256
        const unsigned long long * f2 = (const unsigned long long *) flags;
        unsigned spins = 0; bool ok = true;
        while (true) {
            bool behind = false;
            const unsigned i0 = threadIdx.x * 4;
            if (i0 < n) {
                const unsigned long long a = __hip_atomic_load(f2 + threadIdx.x * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long b = __hip_atomic_load(f2 + threadIdx.x * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                behind = (unsigned) a < epoch || (unsigned)(a >> 32) < epoch || (unsigned) b < epoch || (unsigned)(b >> 32) < epoch;
            }
            if (!__any(behind)) break;
            if (++spins > (1u << 22)) { *abort_flag = 1; ok = false; break; }
        }
        if (threadIdx.x == 0) okflag = ok;
    }
    __syncthreads();
    return okflag != 0;
}

__device__ __forceinline__ bool grid_barrier(unsigned * counter, unsigned target, unsigned * abort_flag, int variant, unsigned it) {
    if (variant == 4) return flag_barrier(counter, it + 1, abort_flag);
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        unsigned spins = 0;
        if (variant == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 22)) { *abort_flag = 1; ok = false; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        } else if (variant == 1 || variant == 2) {
            if (variant == 2) __atomic_thread_fence(__ATOMIC_RELEASE);   // hipcc: agent scope by default for __threadfence(); use the builtin below
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                if (++spins > (1u << 24)) { *abort_flag = 1; ok = false; break; }
            }
            if (variant == 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        } else {
            // counters: [0] global, [16 + 16 * xcd] per XCD (separate cache lines); workgroup b runs on XCD b % 8
            const unsigned xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
            unsigned * local = counter + 16 + 16 * xcd;
            const unsigned prev = __hip_atomic_fetch_add(local, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (prev == (it + 1) * per_xcd - 1) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (it + 1) * 8) {
                if (++spins > (1u << 24)) { *abort_flag = 1; ok = false; break; }
            }
        }
    }
    __syncthreads();
    return ok;
}

// mode 0: barriers only.  mode 1: every workgroup publishes 64 floats, barrier, every workgroup reads all of them (4-byte coherent loads).
// mode 2: a 4096-float vector (16 floats per workgroup), read back with 8-byte agent-coherent loads (2 per thread).
// mode 3: same vector, published with coherent stores, read with PLAIN 16-byte loads after a buffer_inv sc1.
__global__ void __launch_bounds__(1024) k_bar(unsigned * counter, unsigned * abort_flag, int iters, int mode, float * xchg, float * out, int variant) {
    const unsigned nwg = gridDim.x;
    float acc = 0.0f;
    for (int it = 0; it < iters; it++) {
        if (mode == 1) {
            float * slot = xchg + (size_t)(it & 1) * nwg * 64;
            if (threadIdx.x < 64) __hip_atomic_store(slot + blockIdx.x * 64 + threadIdx.x, (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mode >= 2) {
            float * slot = xchg + (size_t)(it & 1) * nwg * 16;
            if (threadIdx.x < 16) __hip_atomic_store(slot + blockIdx.x * 16 + threadIdx.x, (float)(it + blockIdx.x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!grid_barrier(counter, (unsigned)(it + 1) * nwg, abort_flag, variant, (unsigned) it)) return;
        if (mode == 1) {
            const float * slot = xchg + (size_t)(it & 1) * nwg * 64;
            for (unsigned i = threadIdx.x; i < nwg * 64; i += blockDim.x) acc += __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (mode == 2) {
            const unsigned long long * slot = (const unsigned long long *)(xchg + (size_t)(it & 1) * nwg * 16);
            for (unsigned i = threadIdx.x; i < nwg * 8; i += blockDim.x) {
                const unsigned long long u = __hip_atomic_load(slot + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                acc += __uint_as_float((unsigned) u) + __uint_as_float((unsigned)(u >> 32));
            }
        }
        if (mode == 3) {
            asm volatile("buffer_inv sc1" ::: "memory");
            const float4 * slot = (const float4 *)(xchg + (size_t)(it & 1) * nwg * 16);
            for (unsigned i = threadIdx.x; i < nwg * 4; i += blockDim.x) { const float4 u = slot[i]; acc += (u.x + u.y) + (u.z + u.w); }
        }
    }
    if (mode >= 1) atomicAdd(out, acc);
}

int main(int argc, char ** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned * counter, * abort_flag; float * xchg, * out;
    CK(hipMalloc(&counter, 4 * 256)); CK(hipMalloc(&abort_flag, 4)); CK(hipMalloc(&xchg, (size_t) 2 * 1024 * 64 * 4)); CK(hipMalloc(&out, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int variant : {3, 4}) for (int wg : {1024}) for (int per_cu : {1}) for (int mode : {0, 1, 2, 3}) {
        const int grid = cus * per_cu;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(counter, 0, 4 * 256)); CK(hipMemset(abort_flag, 0, 4)); CK(hipMemset(out, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_bar, dim3(grid), dim3(wg), 0, 0, counter, abort_flag, iters, mode, xchg, out, variant);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            unsigned ab; float o; CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&o, out, 4, hipMemcpyDeviceToHost));
            if (rep == 1) {
                // expected sum for mode 1: sum over it, over readers (grid), over (it + b) * 64
                double exp = 0; for (int it = 0; it < iters; it++) for (int b = 0; b < grid; b++) exp += (mode == 1 ? 64.0 : 16.0) * (it + b);
                exp *= grid;
                printf("variant %d grid %4d x %4d thr  mode %d : %.3f us per barrier  (abort %u, sum ratio %.6f)\n", variant, grid, wg, mode, ms * 1000.0 / iters, ab,
                       mode ? (double) o / exp : 1.0);
            }
        }
    }
    return 0;
}
