// tp_oneshot.hip -- one-shot direct-write all-reduce for the tensor-parallel decode messages ([hidden] fp32 = 16-32 KB, twice per layer).
//
// A ring all-reduce over xGMI (RCCL) pays 2 (N - 1) hops of launch-and-wait latency for a message that fits one packet burst; at this size the right shape is
// ONE step: every rank writes its partial vector straight into a slot of every peer's receive buffer (peer memory mapped through HIP IPC: one process per GPU)
// and sums the N slots of its own buffer IN RANK ORDER -- every rank computes the same bits, and for two ranks the same bits as any other sum.  One kernel launch
// per all-reduce, no host round trip, capturable in the decode hipGraph (the sequence number lives in device memory).  The reference has no tensor parallelism
// (SplitMethod::Row is a TODO, src/backend.h:322-327); SURVEY.md 8(e).
//
// Round 5: DATA-TAGGED GRANULES instead of payload + fence + flag (MI355X_MICROARCH.md rows handoff-1to1 / handoff-flag: a separate flag costs 1.7-2.5 x the
// tagged form, and scalar 4-byte system-scope stores are one fabric write each).  A granule is 8 bytes {value, sequence number}; a thread owns four elements, builds
// their four granules and sends them as TWO 16-byte write-through stores (sc0 sc1) per peer; it then polls the same four granules of every rank's slot in its OWN
// buffer with 16-byte sc0 sc1 loads until all tags carry this all-reduce's number and adds the values in rank order.  Thread <-> element is one-to-one on both
// sides, so there is no fence, no flag, no barrier in the kernel, and it runs on as many workgroups as the vector has 1024-element pieces (4 for 4096, 8 for 8192).
//
//   buffer of a rank:  granules [2 parities][nranks][max_n] x 8 bytes
//   all-reduce number s (parity s & 1):  (1) my granules -> slot[par][rank] of every rank's buffer   (2) poll own slot[par][r], all r, tag == s   (3) buf = sum, r ascending
//   A rank can be at most one all-reduce ahead of a peer (it needs the peer's granules of s to finish s), so two parities keep a slot from being overwritten
//   while a slower rank still reads it.  The wait is bounded: on a timeout the error word is raised and the launch returns (no hang).  The sequence number is
//   advanced by the LAST workgroup to finish (every workgroup has read it by then).
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

__device__ __forceinline__ size_t tpo_slot_off(const tpo_args & a, int par, int r) { return ((size_t) par * a.nranks + r) * a.max_n * 8; }
// 16-byte system-scope (write-through / cache-bypassing) store and load: no builtin carries sc0 sc1 on a dwordx4
__device__ __forceinline__ void tpo_store16(void * p, u32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" :: "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ u32x4 tpo_load16(const void * p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }

// seq[0] = all-reduces done, seq[1] = error word, seq[2] = workgroups of the running launch that have finished
__global__ void __launch_bounds__(256) k_tp_oneshot(const tpo_args a, float * __restrict__ buf, int n) {
    const int e = (blockIdx.x * 256 + threadIdx.x) * 4;              // this thread's four elements (n % 4 == 0: the launcher checks)
    const unsigned s = __hip_atomic_load(a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    const int par = (int)(s & 1u);
    if (e < n) {
        const f32x4 mine = *(const f32x4 *)(buf + e);
        const u32x4 g0 = { __float_as_uint(mine.x), s, __float_as_uint(mine.y), s }, g1 = { __float_as_uint(mine.z), s, __float_as_uint(mine.w), s };
        // (1) my four granules into my slot of every rank's buffer (my own included)
        for (int p = 0; p < a.nranks; p++) {
            char * dst = a.peer[p] + tpo_slot_off(a, par, a.rank) + (size_t) e * 8;
            tpo_store16(dst, g0); tpo_store16(dst + 16, g1);
        }
        // (2) + (3) everybody's granules of the same elements in MY buffer, summed in rank order
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int r = 0; r < a.nranks; r++) {
            const char * src = a.peer[a.rank] + tpo_slot_off(a, par, r) + (size_t) e * 8;
            u32x4 h0, h1;
            int spins = 0;
            for (;;) {
                h0 = tpo_load16(src); h1 = tpo_load16(src + 16);
                if (h0.y == s && h0.w == s && h1.y == s && h1.w == s) break;
                __builtin_amdgcn_s_sleep(1);
                // bounded: a peer that never delivers costs ONE time-out, not one per slot and launch -- once the error word is up every later wait gives up at its next look
                if ((++spins & 1023) == 0 && (spins > (1 << 21) || __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
                }
            }
            const f32x4 v = { __uint_as_float(h0.x), __uint_as_float(h0.z), __uint_as_float(h1.x), __uint_as_float(h1.z) };
            if (r == 0) acc = v; else { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
        }
        *(f32x4 *)(buf + e) = acc;
    }
    // the last workgroup to finish advances the sequence number: every workgroup of this launch has read it by then
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.seq + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x - 1) {
            __hip_atomic_store(a.seq + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.seq, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static size_t tpo_bytes(int nranks, size_t max_n) { return ((size_t) 2 * nranks * max_n * 8 + 255) & ~(size_t) 255; }

// rank's receive buffer; handle64 <- the 64-byte IPC handle the other ranks open (exchange it by any host-side means, e.g. torch.distributed.all_gather)
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_create(int rank, int nranks, size_t max_n, void ** out, void * handle64) {
    if (!out || !handle64 || nranks < 1 || nranks > TPO_MAX_RANKS || rank < 0 || rank >= nranks || max_n == 0 || max_n % 4) FAIL(CLLM_E_INVALID, "tp_oneshot_create: arguments");
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
    HIP_TRY(hipMalloc((void **) &o->seq, 16));
    HIP_TRY(hipMemset(o->seq, 0, 16));
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
    if (n > o->max_n || n % 4 || (((uintptr_t) buf) & 15)) return CLLM_E_UNSUPPORTED;      // a prompt-sized (or unaligned) message: the caller takes the ring all-reduce
    for (int r = 0; r < o->nranks; r++) if (!o->peer[r]) FAIL(CLLM_E_INVALID, "tp_oneshot_all_reduce: cllm_tp_oneshot_connect first");
    tpo_args a;
    for (int r = 0; r < TPO_MAX_RANKS; r++) a.peer[r] = r < o->nranks ? o->peer[r] : nullptr;
    a.rank = o->rank; a.nranks = o->nranks; a.max_n = o->max_n; a.seq = o->seq; a.err = o->err;
    hipLaunchKernelGGL(k_tp_oneshot, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, (hipStream_t) stream, a, buf, (int) n);
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
// the error word is sticky.  After the host has reported the time-out (cllm_tp_oneshot_error) and brought every rank to a common boundary (all streams idle, the
// same number of all-reduces issued on every rank), this clears it; the sequence numbers are untouched (they only ever advance together).
extern "C" __attribute__((visibility("default")))
int cllm_tp_oneshot_clear_error(void * os) {
    tp_oneshot * o = (tp_oneshot *) os;
    if (!o || !o->seq) FAIL(CLLM_E_INVALID, "tp_oneshot_clear_error: null");
    HIP_TRY(hipMemset(o->seq + 1, 0, 4));
    return CLLM_OK;
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


// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
// The receive buffers of the all-reduce FUSED into the decode mat-vecs (gemv_tp.hip; kernel forms EPI 4 / PRO 5 of gemv_decode_kernel.h): granules [site][rank][max_n] x 8 bytes
// per rank, peers mapped through HIP IPC exactly as above; the device-side context the kernels read (peer table, rank, step word, error word) lives in device memory.
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------
struct tp_fuse_dev_h { char * peer[16]; int rank, nranks; unsigned max_n, pad; const unsigned * step; unsigned * err; };       // == tp_fuse_dev (gemv_decode_kernel.h)
struct tp_fused {
    int rank, nranks, n_sites; size_t max_n;
    char * local; char * peer[TPO_MAX_RANKS]; bool opened[TPO_MAX_RANKS];
    unsigned * words;                          // device: [0] step number, [1] error word
    tp_fuse_dev_h * dev;                       // device copy of the context
    bool fine_grained;
};
__global__ void k_tpf_advance(unsigned * step) { step[0] = step[0] + 1u; }

extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_create(int rank, int nranks, int n_sites, size_t max_n, void ** out, void * handle64) {
    if (!out || !handle64 || nranks < 1 || nranks > TPO_MAX_RANKS || rank < 0 || rank >= nranks || n_sites < 1 || max_n == 0 || max_n % 4) FAIL(CLLM_E_INVALID, "tp_fused_create: arguments");
    tp_fused * o = new tp_fused();
    o->rank = rank; o->nranks = nranks; o->n_sites = n_sites; o->max_n = max_n;
    const size_t bytes = ((size_t) n_sites * nranks * max_n * 8 + 255) & ~(size_t) 255;
    hipError_t e = hipExtMallocWithFlags((void **) &o->local, bytes, hipDeviceMallocFinegrained);      // see cllm_tp_oneshot_create: polling what a REMOTE GPU writes needs fine-grained memory
    hipIpcMemHandle_t h;
    if (e == hipSuccess) { e = hipIpcGetMemHandle(&h, o->local); if (e != hipSuccess) { (void) hipFree(o->local); o->local = nullptr; } }
    o->fine_grained = e == hipSuccess;
    if (e != hipSuccess) {
        (void) hipGetLastError();
        const char * same = getenv("CLLM_TP_ONESHOT_SAME_DEVICE");
        if (!same || strcmp(same, "1")) { delete o; FAIL(CLLM_E_UNSUPPORTED, "tp_fused_create: fine-grained IPC memory is not available (%s); the coarse-grained fallback is only valid when all ranks share one GPU (CLLM_TP_ONESHOT_SAME_DEVICE=1)", hipGetErrorString(e)); }
        HIP_TRY(hipMalloc((void **) &o->local, bytes));
        HIP_TRY(hipIpcGetMemHandle(&h, o->local));
    }
    HIP_TRY(hipMemset(o->local, 0, bytes));
    HIP_TRY(hipMalloc((void **) &o->words, 16));
    HIP_TRY(hipMemset(o->words, 0, 16));
    HIP_TRY(hipMalloc((void **) &o->dev, sizeof(tp_fuse_dev_h)));
    HIP_TRY(hipDeviceSynchronize());
    o->peer[rank] = o->local;
    memcpy(handle64, &h, 64);
    *out = o;
    return CLLM_OK;
}
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_connect(void * os, const void * handles) {
    tp_fused * o = (tp_fused *) os;
    if (!o || !handles) FAIL(CLLM_E_INVALID, "tp_fused_connect: null");
    for (int r = 0; r < o->nranks; r++) {
        if (r == o->rank) continue;
        hipIpcMemHandle_t h; memcpy(&h, (const char *) handles + 64 * r, 64);
        void * p = nullptr;
        HIP_TRY(hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess));
        o->peer[r] = (char *) p; o->opened[r] = true;
    }
    tp_fuse_dev_h hd;
    memset(&hd, 0, sizeof(hd));
    for (int r = 0; r < o->nranks; r++) hd.peer[r] = o->peer[r];
    hd.rank = o->rank; hd.nranks = o->nranks; hd.max_n = (unsigned) o->max_n; hd.step = o->words; hd.err = o->words + 1;
    HIP_TRY(hipMemcpy(o->dev, &hd, sizeof(hd), hipMemcpyHostToDevice));
    return CLLM_OK;
}
extern "C" int cllm_tp_fused_destroy(void * os);
// All ranks inside ONE process (the ggml module's logical tensor-parallel device, chatllm.cpp_amd/host/ggml-hip.cpp: the reference's host is one process that drives every GPU,
// src/backend.cpp:677-778): rank r's buffers live on devices[r]; peers are plain pointers (peer access is enabled between distinct GPUs), no IPC handles.  Several ranks may share
// a GPU (virtual ranks: the one-GPU test vehicle) -- their launches must then be issued on ONE stream in site order, so that a gather never polls for a launch that cannot start.
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_create_group(int nranks, const int * devices, int n_sites, size_t max_n, void ** out) {
    if (!out || !devices || nranks < 1 || nranks > TPO_MAX_RANKS || n_sites < 1 || max_n == 0 || max_n % 4) FAIL(CLLM_E_INVALID, "tp_fused_create_group: arguments");
    int keep = 0;
    HIP_TRY(hipGetDevice(&keep));
    bool distinct = false;
    for (int r = 1; r < nranks; r++) distinct = distinct || devices[r] != devices[0];
    const size_t bytes = ((size_t) n_sites * nranks * max_n * 8 + 255) & ~(size_t) 255;
    tp_fused * os[TPO_MAX_RANKS] = {};
    auto undo = [&]() { for (int r = 0; r < nranks; r++) if (os[r]) { (void) hipSetDevice(devices[r]); (void) cllm_tp_fused_destroy(os[r]); } (void) hipSetDevice(keep); };
    for (int r = 0; r < nranks; r++) {
        if (hipSetDevice(devices[r]) != hipSuccess) { undo(); FAIL(CLLM_E_HIP, "tp_fused_create_group: device %d", devices[r]); }
        tp_fused * o = new tp_fused();
        os[r] = o;
        o->rank = r; o->nranks = nranks; o->n_sites = n_sites; o->max_n = max_n;
        // what a REMOTE GPU writes and this one polls must be fine-grained; ranks that all share one GPU meet in its L2 (plain memory)
        hipError_t e = distinct ? hipExtMallocWithFlags((void **) &o->local, bytes, hipDeviceMallocFinegrained) : hipMalloc((void **) &o->local, bytes);
        o->fine_grained = distinct && e == hipSuccess;
        if (e == hipSuccess) e = hipMemset(o->local, 0, bytes);
        if (e == hipSuccess) e = hipMalloc((void **) &o->words, 16);
        if (e == hipSuccess) e = hipMemset(o->words, 0, 16);
        if (e == hipSuccess) e = hipMalloc((void **) &o->dev, sizeof(tp_fuse_dev_h));
        if (e != hipSuccess) { (void) hipGetLastError(); undo(); FAIL(CLLM_E_ALLOC, "tp_fused_create_group: rank %d on device %d: %s", r, devices[r], hipGetErrorString(e)); }
        for (int q = 0; q < nranks; q++) if (devices[q] != devices[r]) {
            const hipError_t pe = hipDeviceEnablePeerAccess(devices[q], 0);
            if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { (void) hipGetLastError(); undo(); FAIL(CLLM_E_UNSUPPORTED, "tp_fused_create_group: no peer access from device %d to %d (%s)", devices[r], devices[q], hipGetErrorString(pe)); }
            (void) hipGetLastError();
        }
    }
    for (int r = 0; r < nranks; r++) {
        tp_fuse_dev_h hd;
        memset(&hd, 0, sizeof(hd));
        for (int q = 0; q < nranks; q++) { os[r]->peer[q] = os[q]->local; hd.peer[q] = os[q]->local; }
        hd.rank = r; hd.nranks = nranks; hd.max_n = (unsigned) max_n; hd.step = os[r]->words; hd.err = os[r]->words + 1;
        if (hipSetDevice(devices[r]) != hipSuccess || hipMemcpy(os[r]->dev, &hd, sizeof(hd), hipMemcpyHostToDevice) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { undo(); FAIL(CLLM_E_HIP, "tp_fused_create_group: context of rank %d", r); }
    }
    HIP_TRY(hipSetDevice(keep));
    for (int r = 0; r < nranks; r++) out[r] = os[r];
    return CLLM_OK;
}
// the device-side context (kernel argument of the EPI 4 / PRO 5 launches), the number of sites and the vector length the buffers hold
extern "C" __attribute__((visibility("default"))) const void * cllm_tp_fused_dev(void * os) { return os ? (const void *) ((tp_fused *) os)->dev : nullptr; }
extern "C" __attribute__((visibility("default"))) int cllm_tp_fused_sites(void * os) { return os ? ((tp_fused *) os)->n_sites : 0; }
extern "C" __attribute__((visibility("default"))) size_t cllm_tp_fused_max_n(void * os) { return os ? ((tp_fused *) os)->max_n : 0; }
extern "C" __attribute__((visibility("default"))) int cllm_tp_fused_fine_grained(void * os) { return os ? (int) ((tp_fused *) os)->fine_grained : -1; }
// one step = one pass over the sites: every rank calls it once at the start of every decode step, stream-ordered (capturable)
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_advance(void * os, void * stream) {
    tp_fused * o = (tp_fused *) os;
    if (!o) FAIL(CLLM_E_INVALID, "tp_fused_advance: null");
    hipLaunchKernelGGL(k_tpf_advance, dim3(1), dim3(1), 0, (hipStream_t) stream, o->words);
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_error(void * os) {
    tp_fused * o = (tp_fused *) os;
    if (!o) return 1;
    unsigned e = 0;
    if (hipMemcpy(&e, o->words + 1, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    return (int) e;
}
// the error word was sticky with no way back.  After the host has reported the time-out (cllm_tp_fused_error) and brought every rank to a common step boundary
// (all streams idle), this clears it; the step numbers are untouched (they only ever advance together).
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_clear_error(void * os) {
    tp_fused * o = (tp_fused *) os;
    if (!o) FAIL(CLLM_E_INVALID, "tp_fused_clear_error: null");
    HIP_TRY(hipMemset(o->words + 1, 0, 4));
    return CLLM_OK;
}
extern "C" __attribute__((visibility("default")))
int cllm_tp_fused_destroy(void * os) {
    tp_fused * o = (tp_fused *) os;
    if (!o) return CLLM_OK;
    for (int r = 0; r < o->nranks; r++) if (o->opened[r] && o->peer[r]) (void) hipIpcCloseMemHandle(o->peer[r]);
    if (o->local) (void) hipFree(o->local);
    if (o->words) (void) hipFree(o->words);
    if (o->dev) (void) hipFree(o->dev);
    delete o;
    return CLLM_OK;
}
