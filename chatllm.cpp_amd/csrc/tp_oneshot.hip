// tp_oneshot.hip -- one-shot direct-write all-reduce for the tensor-parallel decode messages ([hidden] fp32 = 16-32 KB, twice per layer).
//
// A ring all-reduce over xGMI (RCCL) pays 2 (N - 1) hops of launch-and-wait latency for a message that fits one packet burst; at this size the right shape is
// ONE step: every rank writes its partial vector straight into a slot of every peer's receive buffer (peer memory mapped through HIP IPC: one process per GPU),
// raises a flag there, waits until its own buffer holds everybody's flag, and sums the N slots IN RANK ORDER -- every rank computes the same bits, and for two
// ranks the same bits as any other sum.  One kernel launch per all-reduce, no host round trip, capturable in the decode hipGraph (the sequence number lives in
// device memory).  The reference has no tensor parallelism (SplitMethod::Row is a TODO, src/backend.h:322-327); SURVEY.md 8(e).
//
//   buffer of a rank:  slots [2 parities][nranks][max_n] floats | flags [2 parities][nranks] uint32
//   all-reduce number s (parity s & 1):  (1) buf -> slot[par][rank] of every rank's buffer   (2) system fence, flag[par][rank] = s everywhere
//                                        (3) wait own flag[par][r] == s for all r              (4) buf = sum_r slot[par][r] (own buffer), r ascending
//   A rank can be at most one all-reduce ahead of a peer (it needs the peer's flag of s to finish s), so two parities keep a slot from being overwritten
//   while a slower rank still reads it.  The wait is bounded: on a timeout the error word is raised and the launch returns (no hang).
#include "common.h"
#include <string.h>
#include <stdlib.h>

#define TPO_MAX_RANKS 16

struct tp_oneshot {
    int rank, nranks; size_t max_n;
    char * local;
    char * peer[TPO_MAX_RANKS];
    unsigned * seq;                            // device: all-reduces done so far
    unsigned * err;                            // device: set when a wait timed out
    bool opened[TPO_MAX_RANKS];
    bool fine_grained;                         // the receive buffer is fine-grained (coherent across agents while a kernel runs)
};
struct tpo_args { char * peer[TPO_MAX_RANKS]; int rank, nranks; unsigned long long max_n; unsigned * seq, * err; };

__device__ __forceinline__ size_t tpo_slot_off(const tpo_args & a, int par, int r) { return ((size_t) par * a.nranks + r) * a.max_n * 4; }
__device__ __forceinline__ size_t tpo_flag_off(const tpo_args & a, int par, int r) { return (size_t) 2 * a.nranks * a.max_n * 4 + ((size_t) par * a.nranks + r) * 4; }

__global__ void __launch_bounds__(1024) k_tp_oneshot(const tpo_args a, float * __restrict__ buf, int n) {
    const int tid = threadIdx.x;
    const unsigned s = __hip_atomic_load(a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int par = (int)(s & 1u);
    // (1) my partial vector into my slot of every rank's buffer (my own included): system-scope stores, write-through
    for (int p = 0; p < a.nranks; p++) {
        float * dst = (float *)(a.peer[p] + tpo_slot_off(a, par, a.rank));
        for (int i = tid; i < n; i += 1024) __hip_atomic_store(dst + i, buf[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");                 // system scope: the data is visible before the flag
    __syncthreads();
    // (2) the flags
    if (tid < a.nranks) __hip_atomic_store((unsigned *)(a.peer[tid] + tpo_flag_off(a, par, a.rank)), s, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // (3) everybody's flag in MY buffer
    if (tid < a.nranks) {
        const unsigned * f = (const unsigned *)(a.peer[a.rank] + tpo_flag_off(a, par, tid));
        int spins = 0;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != s) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1 << 22)) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    // (4) the sum, rank order
    for (int i = tid; i < n; i += 1024) {
        float v = __hip_atomic_load((const float *)(a.peer[a.rank] + tpo_slot_off(a, par, 0)) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int r = 1; r < a.nranks; r++) v = v + __hip_atomic_load((const float *)(a.peer[a.rank] + tpo_slot_off(a, par, r)) + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        buf[i] = v;
    }
    if (tid == 0) __hip_atomic_store(a.seq, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

static size_t tpo_bytes(int nranks, size_t max_n) { return ((size_t) 2 * nranks * max_n * 4 + (size_t) 2 * nranks * 4 + 255) & ~(size_t) 255; }

// rank's receive buffer; handle64 <- the 64-byte IPC handle the other ranks open (exchange it by any host-side means, e.g. torch.distributed.all_gather)
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_create(int rank, int nranks, size_t max_n, void ** out, void * handle64) {
    if (!out || !handle64 || nranks < 1 || nranks > TPO_MAX_RANKS || rank < 0 || rank >= nranks || max_n == 0) FAIL(CLLM_E_INVALID, "tp_oneshot_create: arguments");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
    tp_oneshot * o = new tp_oneshot();
    o->rank = rank; o->nranks = nranks; o->max_n = max_n;
    const size_t bytes = tpo_bytes(nranks, max_n);
    // fine-grained memory (coherent across agents while a kernel runs) is what makes polling a flag that a REMOTE GPU writes over xGMI valid.  Plain (coarse-grained)
    // device memory is coherent only inside one GPU (through its L2): accepted only when the caller states that every rank runs on this same device
    // (CLLM_TP_ONESHOT_SAME_DEVICE=1: the two-processes-on-one-GPU tests) -- never silently, a stale flag would show up as wrong tokens, not as an error.
    hipError_t e = hipExtMallocWithFlags((void **) &o->local, bytes, hipDeviceMallocFinegrained);
    hipIpcMemHandle_t h;
    if (e == hipSuccess) { e = hipIpcGetMemHandle(&h, o->local); if (e != hipSuccess) { (void) hipFree(o->local); o->local = nullptr; } }
    o->fine_grained = e == hipSuccess;
    if (e != hipSuccess) {
        (void) hipGetLastError();
        const char * same = getenv("CLLM_TP_ONESHOT_SAME_DEVICE");
        if (!same || strcmp(same, "1")) { delete o; FAIL(CLLM_E_UNSUPPORTED, "tp_oneshot_create: fine-grained IPC memory is not available (%s); the coarse-grained fallback is only valid when all ranks share one GPU (CLLM_TP_ONESHOT_SAME_DEVICE=1)", hipGetErrorString(e)); }
        HIP_TRY(hipMalloc((void **) &o->local, bytes));
        HIP_TRY(hipIpcGetMemHandle(&h, o->local));
    }
    HIP_TRY(hipMemset(o->local, 0, bytes));
    HIP_TRY(hipMalloc((void **) &o->seq, 8));
    HIP_TRY(hipMemset(o->seq, 0, 8));
    o->err = o->seq + 1;
    HIP_TRY(hipDeviceSynchronize());
    o->peer[rank] = o->local;
    memcpy(handle64, &h, 64);
    *out = o;
    return CLLM_OK;
}
// handles: nranks x 64 bytes, rank-ordered (this rank's own entry is ignored)
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_connect(void * os, const void * handles) {
    tp_oneshot * o = (tp_oneshot *) os;
    if (!o || !handles) FAIL(CLLM_E_INVALID, "tp_oneshot_connect: null");
    for (int r = 0; r < o->nranks; r++) {
        if (r == o->rank) continue;
        hipIpcMemHandle_t h; memcpy(&h, (const char *) handles + 64 * r, 64);
        void * p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        o->peer[r] = (char *) p; o->opened[r] = true;
    }
    return CLLM_OK;
}
// in-place sum of n floats over the group, stream-ordered on `stream` (one launch; capturable in a hipGraph); every rank must call it the same number of times
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_all_reduce_f32(void * os, void * stream, float * buf, size_t n) {
    tp_oneshot * o = (tp_oneshot *) os;
    if (!o || !buf || n == 0) FAIL(CLLM_E_INVALID, "tp_oneshot_all_reduce: null");
    if (n > o->max_n) return CLLM_E_UNSUPPORTED;                   // a prompt-sized message: the caller takes the ring all-reduce
    for (int r = 0; r < o->nranks; r++) if (!o->peer[r]) FAIL(CLLM_E_INVALID, "tp_oneshot_all_reduce: cllm_tp_oneshot_connect first");
    tpo_args a;
    for (int r = 0; r < TPO_MAX_RANKS; r++) a.peer[r] = r < o->nranks ? o->peer[r] : nullptr;
    a.rank = o->rank; a.nranks = o->nranks; a.max_n = o->max_n; a.seq = o->seq; a.err = o->err;
    hipLaunchKernelGGL(k_tp_oneshot, dim3(1), dim3(1024), 0, (hipStream_t) stream, a, buf, (int) n);
    LAUNCH_CHECK();
    return CLLM_OK;
}
// 1: the receive buffer is fine-grained memory, 0: coarse-grained (ranks sharing one GPU, CLLM_TP_ONESHOT_SAME_DEVICE=1), -1: null
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_fine_grained(void * os) { return os ? (int) ((tp_oneshot *) os)->fine_grained : -1; }
// 1 if a wait timed out since creation (a peer died or the calls went out of step)
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_error(void * os) {
    tp_oneshot * o = (tp_oneshot *) os;
    if (!o) return 1;
    unsigned e = 0;
    if (hipMemcpy(&e, o->err, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return (int) e;
}
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_destroy(void * os) {
    tp_oneshot * o = (tp_oneshot *) os;
    if (!o) return CLLM_OK;
    for (int r = 0; r < o->nranks; r++) if (o->opened[r] && o->peer[r]) (void) hipIpcCloseMemHandle(o->peer[r]);
    if (o->local) (void) hipFree(o->local);
    if (o->seq) (void) hipFree(o->seq);
    delete o;
    return CLLM_OK;
}
