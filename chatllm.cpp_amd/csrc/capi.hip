// capi.hip -- the extern "C" surface of libchatllm_hip.so that is not a kernel launcher by itself:
// library/device queries, memory + stream plumbing, and the MUL_MAT / MUL_MAT_ID dispatch.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <tuple>
#include <vector>

// ---- errors -----------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void cllm_set_error(const char * fmt, ...) {
    va_list ap; va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int cllm_hip_check(hipError_t e, const char * what, const char * file, int line) {
    if (e == hipSuccess) return CLLM_OK;
    cllm_set_error("HIP error %d (%s) at %s:%d: %s", (int) e, hipGetErrorString(e), file, line, what);
    (void) hipGetLastError();   // clear sticky launch errors
    return e == hipErrorOutOfMemory ? CLLM_E_ALLOC : CLLM_E_HIP;
}
extern "C" const char * cllm_last_error(void) { return g_err; }
extern "C" int cllm_abi_version(void) { return 1; }

// ---- type traits (ggml.c type_traits[]) -------------------------------------------------------------------
extern "C" size_t cllm_type_size(int type) {
    switch (type) {
        case CLLM_TYPE_F32: case CLLM_TYPE_I32: return 4;
        case CLLM_TYPE_F16: return 2;
        case CLLM_TYPE_I64: return 8;
        case CLLM_TYPE_Q4_0: return 18; case CLLM_TYPE_Q4_1: return 20; case CLLM_TYPE_Q8_0: return 34; case CLLM_TYPE_Q4_K: return 144;
        case CLLM_TYPE_Q5_K: return 176; case CLLM_TYPE_Q6_K: return 210;
        case CLLM_TYPE_Q5_0: return 22; case CLLM_TYPE_Q5_1: return 24; case CLLM_TYPE_IQ4_NL: return 18; case CLLM_TYPE_MXFP4: return 17;
        case CLLM_TYPE_Q2_K: return 84; case CLLM_TYPE_Q3_K: return 110; case CLLM_TYPE_IQ4_XS: return 136; case CLLM_TYPE_TQ1_0: return 54; case CLLM_TYPE_TQ2_0: return 66;
        case CLLM_TYPE_IQ2_XXS: return 66; case CLLM_TYPE_IQ2_XS: return 74; case CLLM_TYPE_IQ2_S: return 82; case CLLM_TYPE_IQ3_XXS: return 98; case CLLM_TYPE_IQ3_S: return 110; case CLLM_TYPE_IQ1_S: return 50; case CLLM_TYPE_IQ1_M: return 56;
    }
    return 0;
}
extern "C" int cllm_blck_size(int type) {
    switch (type) {
        case CLLM_TYPE_F32: case CLLM_TYPE_I32: case CLLM_TYPE_F16: case CLLM_TYPE_I64: return 1;
        case CLLM_TYPE_Q4_0: case CLLM_TYPE_Q4_1: case CLLM_TYPE_Q8_0: case CLLM_TYPE_Q5_0: case CLLM_TYPE_Q5_1: case CLLM_TYPE_IQ4_NL: case CLLM_TYPE_MXFP4: return 32;
        case CLLM_TYPE_Q4_K: case CLLM_TYPE_Q5_K: case CLLM_TYPE_Q6_K: case CLLM_TYPE_Q2_K: case CLLM_TYPE_Q3_K: case CLLM_TYPE_IQ4_XS: case CLLM_TYPE_TQ1_0: case CLLM_TYPE_TQ2_0:
        case CLLM_TYPE_IQ2_XXS: case CLLM_TYPE_IQ2_XS: case CLLM_TYPE_IQ2_S: case CLLM_TYPE_IQ3_XXS: case CLLM_TYPE_IQ3_S: case CLLM_TYPE_IQ1_S: case CLLM_TYPE_IQ1_M: return 256;
    }
    return 0;
}
extern "C" size_t cllm_row_size(int type, int64_t ne) {
    const int bs = cllm_blck_size(type);
    return bs ? cllm_type_size(type) * (size_t)(ne / bs) : 0;
}

// ---- device -----------------------------------------------------------------------------------------------
extern "C" int cllm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void) hipGetLastError(); return 0; }
    return n;
}
extern "C" int cllm_set_device(int device) { HIP_TRY(hipSetDevice(device)); return CLLM_OK; }

int device_cu_count() {
    static thread_local int dev_cached = -1, cus = 256;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return cus;
    if (dev != dev_cached) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
        dev_cached = dev;
    }
    return cus;
}
extern "C" int cllm_device_info(int device, char * name, size_t name_len, size_t * mem_free, size_t * mem_total, int * n_cu) {
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len) { snprintf(name, name_len, "%s (%s)", p.name, p.gcnArchName); }
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (mem_free || mem_total) {
        int cur = 0; HIP_TRY(hipGetDevice(&cur));
        HIP_TRY(hipSetDevice(device));
        size_t f = 0, t = 0; HIP_TRY(hipMemGetInfo(&f, &t));
        HIP_TRY(hipSetDevice(cur));
        if (mem_free) *mem_free = f;
        if (mem_total) *mem_total = t;
    }
    return CLLM_OK;
}

// ---- memory / streams ------------------------------------------------------------------------------------------
extern "C" int cllm_malloc(void ** ptr, size_t size) {
    if (!ptr) FAIL(CLLM_E_INVALID, "malloc: null");
    *ptr = nullptr;
    if (size == 0) return CLLM_OK;
    HIP_TRY(hipMalloc(ptr, size));
    return CLLM_OK;
}
extern "C" int cllm_free(void * ptr) { if (ptr) HIP_TRY(hipFree(ptr)); return CLLM_OK; }
extern "C" int cllm_memset(void * dst, int value, size_t size, void * stream) {
    if (size) HIP_TRY(hipMemsetAsync(dst, value, size, (hipStream_t) stream));
    return CLLM_OK;
}
extern "C" int cllm_memcpy_h2d(void * dst, const void * src, size_t size, void * stream) {
    if (size) HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyHostToDevice, (hipStream_t) stream));
    return CLLM_OK;
}
extern "C" int cllm_memcpy_d2h(void * dst, const void * src, size_t size, void * stream) {
    if (size) { HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToHost, (hipStream_t) stream)); HIP_TRY(hipStreamSynchronize((hipStream_t) stream)); }
    return CLLM_OK;
}
// page-locked host memory: the staging area of a host binding's get_tensor (a D2H copy into pageable memory is several times slower)
extern "C" int cllm_host_malloc(void ** ptr, size_t size) {
    if (!ptr) FAIL(CLLM_E_INVALID, "host_malloc: null");
    *ptr = nullptr;
    if (hipHostMalloc(ptr, size ? size : 1, hipHostMallocDefault) != hipSuccess) { (void) hipGetLastError(); FAIL(CLLM_E_ALLOC, "host_malloc: %zu bytes", size); }
    return CLLM_OK;
}
extern "C" int cllm_host_free(void * ptr) { if (ptr) HIP_TRY(hipHostFree(ptr)); return CLLM_OK; }
extern "C" int cllm_memcpy_d2d(void * dst, const void * src, size_t size, void * stream) {
    if (size) HIP_TRY(hipMemcpyAsync(dst, src, size, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return CLLM_OK;
}
extern "C" int cllm_stream_create(void ** stream) {
    if (!stream) FAIL(CLLM_E_INVALID, "stream_create: null");
    hipStream_t s; HIP_TRY(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *) s;
    return CLLM_OK;
}
// ---- library-owned scratch, per (device, stream, kind); see common.h ----
namespace {
struct scratch_entry { void * p = nullptr; size_t bytes = 0; std::vector<void *> outgrown; };
std::mutex g_scratch_m;
std::map<std::tuple<int, hipStream_t, int>, scratch_entry> g_scratch;
}
void * stream_scratch(hipStream_t st, int kind, size_t need) {
    int dev = 0; (void) hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_scratch_m);
    scratch_entry & e = g_scratch[std::make_tuple(dev, st, kind)];
    if (need <= e.bytes) return e.p;
    // geometric growth (at least twice the block it replaces, whole 2 MiB pages): a stream that sees prompts / contexts of slowly increasing size keeps
    // log2(max / first) outgrown blocks whose sizes sum to less than the live one -- not one dead block per growth step
    size_t want = need > 2 * e.bytes ? need : 2 * e.bytes;
    want = (want + ((size_t) 2 << 20) - 1) & ~(((size_t) 2 << 20) - 1);
    void * np = nullptr;
    if (hipMalloc(&np, want) != hipSuccess) {
        (void) hipGetLastError();
        want = need;                                   // the doubled size does not fit: the exact one may
        if (hipMalloc(&np, want) != hipSuccess) { (void) hipGetLastError(); cllm_set_error("scratch: hipMalloc(%zu) failed", need); return nullptr; }
    }
    if (e.p) e.outgrown.push_back(e.p);               // a captured launch may still address it: kept until the stream goes
    e.p = np; e.bytes = want;
    return np;
}
// every block of `st`, whatever device is current now (the key's device is where the block lives; hipFree takes any device's pointer).  The NULL stream handle is the
// same on every device: only the CURRENT device's entries go then (the caller vouches for that device being idle, not for the others)
void stream_scratch_release(hipStream_t st) {
    int dev = 0; (void) hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(g_scratch_m);
    for (auto it = g_scratch.begin(); it != g_scratch.end(); ) {
        if (std::get<1>(it->first) == st && (st != nullptr || std::get<0>(it->first) == dev)) {
            if (it->second.p) (void) hipFree(it->second.p);
            for (void * q : it->second.outgrown) (void) hipFree(q);
            it = g_scratch.erase(it);
        } else ++it;
    }
}
// scratch of streams the library did not create (the null stream, a host's own): cllm_stream_destroy never sees them.  Call with the device idle.
extern "C" CLLM_API int cllm_scratch_release(void * stream) { stream_scratch_release((hipStream_t) stream); return CLLM_OK; }
extern "C" int cllm_stream_destroy(void * stream) {
    if (stream) { HIP_TRY(hipStreamSynchronize((hipStream_t) stream)); stream_scratch_release((hipStream_t) stream); HIP_TRY(hipStreamDestroy((hipStream_t) stream)); }
    return CLLM_OK;
}
extern "C" int cllm_stream_sync(void * stream) { HIP_TRY(hipStreamSynchronize((hipStream_t) stream)); return CLLM_OK; }
// after a synchronize: did a bounded wait inside a kernel of the current device time out since the last call (gemv_team32.hip's hand-offs between the waves of a team)?
// Such a launch winds down instead of hanging and its results are void: CLLM_E_HIP + cllm_last_error.  A host read of a mapped word: free on the per-token path.
extern "C" int cllm_check_kernel_errors(void) { return gemv_team32_check(); }

// ---- capture / replay of a launch sequence (what ggml_backend_i.graph_plan_create / graph_plan_compute are for, ggml-backend-impl.h:104-113) ----
// Everything launched on `stream` between begin and end becomes one executable graph; replaying it costs one host call and removes the
// per-launch gaps on the GPU.  end returns CLLM_E_UNSUPPORTED (and *graph_exec = NULL) if something in the sequence could not be captured.
extern "C" int cllm_graph_capture_begin(void * stream) {
    if (!stream) FAIL(CLLM_E_INVALID, "graph_capture_begin: the null stream cannot be captured");
    HIP_TRY(hipStreamBeginCapture((hipStream_t) stream, hipStreamCaptureModeRelaxed));
    return CLLM_OK;
}
extern "C" int cllm_graph_capture_end(void * stream, void ** graph_exec) {
    if (!stream || !graph_exec) FAIL(CLLM_E_INVALID, "graph_capture_end: arguments");
    *graph_exec = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t) stream, &graph);
    if (e == hipSuccess) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void) hipGraphDestroy(graph);
    if (e != hipSuccess) { (void) hipGetLastError(); FAIL(CLLM_E_UNSUPPORTED, "graph_capture_end: %s", hipGetErrorString(e)); }
    *graph_exec = (void *) exec;
    return CLLM_OK;
}
extern "C" int cllm_graph_launch(void * graph_exec, void * stream) {
    if (!graph_exec) FAIL(CLLM_E_INVALID, "graph_launch: null");
    HIP_TRY(hipGraphLaunch((hipGraphExec_t) graph_exec, (hipStream_t) stream));
    return CLLM_OK;
}
extern "C" int cllm_graph_destroy(void * graph_exec) { if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t) graph_exec)); return CLLM_OK; }

extern "C" int cllm_event_create(void ** event) {
    if (!event) FAIL(CLLM_E_INVALID, "event_create: null");
    hipEvent_t e; HIP_TRY(hipEventCreate(&e));
    *event = (void *) e;
    return CLLM_OK;
}
extern "C" int cllm_event_destroy(void * event) { if (event) HIP_TRY(hipEventDestroy((hipEvent_t) event)); return CLLM_OK; }
extern "C" int cllm_event_record(void * event, void * stream) { HIP_TRY(hipEventRecord((hipEvent_t) event, (hipStream_t) stream)); return CLLM_OK; }
extern "C" int cllm_event_sync(void * event) { HIP_TRY(hipEventSynchronize((hipEvent_t) event)); return CLLM_OK; }
extern "C" int cllm_stream_wait_event(void * stream, void * event) {
    if (!event) FAIL(CLLM_E_INVALID, "stream_wait_event: null event");
    HIP_TRY(hipStreamWaitEvent((hipStream_t) stream, (hipEvent_t) event, 0));
    return CLLM_OK;
}
extern "C" int cllm_memcpy_peer_async(void * dst, int dst_device, const void * src, int src_device, size_t bytes, void * stream) {
    if (!bytes) return CLLM_OK;
    if (!dst || !src) FAIL(CLLM_E_INVALID, "memcpy_peer_async: null");
    if (dst_device == src_device) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    else                          HIP_TRY(hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, (hipStream_t) stream));
    return CLLM_OK;
}
extern "C" int cllm_event_elapsed_ms(void * start, void * stop, float * ms) {
    if (!ms) FAIL(CLLM_E_INVALID, "event_elapsed: null");
    HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t) start, (hipEvent_t) stop));
    return CLLM_OK;
}

// ---- MUL_MAT dispatch ---------------------------------------------------------------------------------------------
static bool is_quant(int t) { return is_quant_type(t) || is_kq_type(t); }
static int  act_kind(int wtype) { return act_kind_of(wtype); }

// Columns from which the FAST / F16 prefill modes take a quantized product (mmq.hip: exact integer block sums, its own fp32 summation order = tolerance tier;
// dense_f16.hip).  Below it every mode computes in the reference's AVX2 order, so short prompts (BASELINE cfg2: 16 tokens) stay BIT-IDENTICAL to the CPU path end to
// end whatever the mode.  Default 33 (CLLM_MMQ_MIN_COLS).
int mmq_min_cols_get();
static int mmq_min_cols() {
    static int v = -1;
    if (v < 0) { v = 33; if (const char * e = getenv("CLLM_MMQ_MIN_COLS")) { int x = atoi(e); if (x >= 1) v = x; } }
    return v;
}
// Columns from which the exact-order GEMM (mmx.hip: one launch, the weights read once, a 64-token tile whatever the count) beats the exact-order mat-vec in chunks of
// <= 4 columns (mmvq.hip: the weights re-read per chunk) -- the same bits either way.  Measured on Llama-3-8B shapes (profiles/r03_short_prompt_crossover.txt): the
// wide projections (gate/up) cross at 4-5 columns, the hidden-sized ones (qkv, o, down) at 8-11.  CLLM_MMX_MIN_COLS overrides both.
static int exact_gemm_min_cols(int64_t nrows) {
    static int v = -2;
    if (v == -2) { v = -1; if (const char * e = getenv("CLLM_MMX_MIN_COLS")) { int x = atoi(e); if (x >= 1) v = x; } }
    return v > 0 ? v : nrows >= 16384 ? 5 : 10;
}
// 0: mat-vec in column chunks (mmvq), 1: exact-order GEMM (mmx), 2: the fast / f16 mode's kernels
static int gemm_path(int64_t M, int64_t nrows) {
    if (M >= mmq_min_cols() && (prefill_f16_enabled() || prefill_mode() == 0)) return 2;
    return M >= exact_gemm_min_cols(nrows) ? 1 : 0;
}

// ---- prefill mode (common.h) ----
static int g_prefill_mode = -1;
int prefill_mode() {
    if (g_prefill_mode < 0) {
        const char * e = getenv("CLLM_PREFILL");
        g_prefill_mode = (e && (!strcmp(e, "fast") || !strcmp(e, "f16"))) ? 0 : 1;
    }
    return g_prefill_mode;
}
extern "C" int cllm_set_prefill_mode(int mode) {
    if (mode != 0 && mode != 1) FAIL(CLLM_E_INVALID, "set_prefill_mode: 0 (fast) or 1 (exact)");
    g_prefill_mode = mode;
    return CLLM_OK;
}
extern "C" int cllm_get_prefill_mode(void) { return prefill_mode(); }
// the prompt's ATTENTION block on its own switch (round 6: which half of the fast mode owns its deviation from the CPU run -- profiles/r06_prefill_mode_decomposition.txt):
//   CLLM_PREFILL_ATTN=exact | fast overrides what CLLM_PREFILL / cllm_set_prefill_mode say for K.Q / soft_max / V.P; unset (-1): follows prefill_mode()
static int g_prefill_attn_mode = -2;
int prefill_attn_mode() {
    if (g_prefill_attn_mode == -2) {
        const char * e = getenv("CLLM_PREFILL_ATTN");
        g_prefill_attn_mode = !e ? -1 : !strcmp(e, "fast") ? 0 : !strcmp(e, "exact") ? 1 : -1;
    }
    return g_prefill_attn_mode >= 0 ? g_prefill_attn_mode : prefill_mode();
}
extern "C" int cllm_set_prefill_attn_mode(int mode) {
    if (mode < -1 || mode > 1) FAIL(CLLM_E_INVALID, "set_prefill_attn_mode: -1 (follow the prefill mode), 0 (flash kernel) or 1 (exact order)");
    g_prefill_attn_mode = mode;
    return CLLM_OK;
}

extern "C" size_t cllm_mul_mat_wsize(const cllm_tensor * src0, const cllm_tensor * src1) {
    if (!src0 || !src1 || !is_quant(src0->type)) return 0;
    return act_row_bytes(src1->ne[0], act_kind(src0->type)) * (size_t) t_nrows(src1);
}

static int check_mm(const cllm_tensor * src0, const cllm_tensor * src1, const cllm_tensor * dst, const char * name) {
    if (!src0 || !src1 || !dst) FAIL(CLLM_E_INVALID, "%s: null tensor", name);
    if (src1->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32) FAIL(CLLM_E_UNSUPPORTED, "%s: src1/dst must be F32", name);
    if (!is_quant(src0->type) && src0->type != CLLM_TYPE_F16 && src0->type != CLLM_TYPE_F32) FAIL(CLLM_E_UNSUPPORTED, "%s: src0 type %d", name, src0->type);
    if (src0->ne[0] != src1->ne[0]) FAIL(CLLM_E_INVALID, "%s: K mismatch %lld vs %lld", name, (long long) src0->ne[0], (long long) src1->ne[0]);
    if (src0->nb[0] != cllm_type_size(src0->type) || src1->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_UNSUPPORTED, "%s: rows must be dense", name);
    if (src0->ne[0] % cllm_blck_size(src0->type)) FAIL(CLLM_E_INVALID, "%s: K not a multiple of the block size", name);
    if (src0->type == CLLM_TYPE_Q4_1 && ((uintptr_t) src0->data % 4 || src0->nb[1] % 4 || src0->nb[2] % 4 || src0->nb[3] % 4)) FAIL(CLLM_E_UNSUPPORTED, "%s: Q4_1 rows must be 4-byte aligned", name);
    return CLLM_OK;
}

extern "C" int cllm_op_mul_mat(void * stream, const cllm_tensor * src0, const cllm_tensor * src1, cllm_tensor * dst, void * wdata, size_t wsize) {
    int rc = check_mm(src0, src1, dst, "mul_mat");
    if (rc) return rc;
    if (dst->ne[0] != src0->ne[1] || dst->ne[1] != src1->ne[1] || dst->ne[2] != src1->ne[2] || dst->ne[3] != src1->ne[3]) FAIL(CLLM_E_INVALID, "mul_mat: dst shape");
    if (src0->ne[2] <= 0 || src0->ne[3] <= 0 || src1->ne[2] % src0->ne[2] || src1->ne[3] % src0->ne[3]) FAIL(CLLM_E_INVALID, "mul_mat: broadcast");
    if (t_nelements(dst) == 0) return CLLM_OK;
    hipStream_t st = (hipStream_t) stream;
    if (src0->ne[0] == 0) return cllm_memset(dst->data, 0, dst->nb[3] * (size_t) dst->ne[3], stream);

    if (!is_quant(src0->type)) return launch_mul_mat_f(st, src0->type, tv(src0), tv(src1), tv(dst));
    if (is_kq_type(src0->type)) {                      // the coverage types: the exact-order kernel for any number of columns (gemv_kq.hip)
        const size_t stride = act_row_bytes(src0->ne[0], act_kind(src0->type)), need = stride * (size_t) t_nrows(src1);
        if (!wdata || wsize < need) FAIL(CLLM_E_INVALID, "mul_mat: wdata too small (%zu < %zu)", wsize, need);
        if ((uintptr_t) wdata % 16 || (uintptr_t) src1->data % 16 || src1->nb[1] % 16 || src1->nb[2] % 16 || src1->nb[3] % 16) FAIL(CLLM_E_UNSUPPORTED, "mul_mat: operand alignment");
        if ((rc = launch_quantize_act(st, act_kind(src0->type), tv(src1), wdata, stride))) return rc;
        const int64_t r2 = src1->ne[2] / src0->ne[2], r3 = src1->ne[3] / src0->ne[3];
        for (int64_t i13 = 0; i13 < src1->ne[3]; i13++)
        for (int64_t i12 = 0; i12 < src1->ne[2]; i12++) {
            tview w = tv(src0);
            w.data += (i12 / r2) * w.nb[2] + (i13 / r3) * w.nb[3];
            const char * act = (const char *) wdata + (size_t)(i12 * src1->ne[1] + i13 * src1->ne[1] * src1->ne[2]) * stride;
            if (dst->nb[1] % 4) FAIL(CLLM_E_INVALID, "mul_mat: dst stride");
            rc = launch_gemv_kq(st, src0->type, w, act, stride, src1->ne[1], (float *)((char *) dst->data + i12 * dst->nb[2] + i13 * dst->nb[3]), (int64_t)(dst->nb[1] / 4));
            if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "mul_mat: type %d: shape or alignment not taken", src0->type);
            if (rc) return rc;
        }
        return CLLM_OK;
    }

    const int kind = act_kind(src0->type);
    const int64_t K = src0->ne[0];
    const size_t stride = act_row_bytes(K, kind);
    const size_t need = stride * (size_t) t_nrows(src1);
    if (!wdata || wsize < need) FAIL(CLLM_E_INVALID, "mul_mat: wdata too small (%zu < %zu)", wsize, need);
    if ((uintptr_t) wdata % 16 || (uintptr_t) src0->data % 2 || (uintptr_t) src1->data % 16 || src1->nb[1] % 16 || src1->nb[2] % 16 || src1->nb[3] % 16)
        FAIL(CLLM_E_UNSUPPORTED, "mul_mat: operand alignment");
    if (src0->type == CLLM_TYPE_Q4_K && ((uintptr_t) src0->data % 16 || src0->nb[1] % 16 || src0->nb[2] % 16 || src0->nb[3] % 16))
        FAIL(CLLM_E_UNSUPPORTED, "mul_mat: Q4_K rows must be 16-byte aligned");
    // a single dense activation column (decode): quantize it inside the mat-vec (one launch instead of two; the decode kernels)
    if (src1->ne[1] == 1 && src1->ne[2] == 1 && src1->ne[3] == 1 && src0->ne[2] == 1 && src0->ne[3] == 1 && src1->nb[0] == 4 && dst->nb[0] == 4 &&
        src0->nb[1] == cllm_row_size(src0->type, K) && K % act_blk(kind) == 0 && act_row_bytes(K, kind) <= 160 * 1024) {
        rc = launch_gemv_decode(st, src0->type, src0->data, K, src0->ne[1], 2, (const float *) src1->data, nullptr, 0.0f, 0, (float *) dst->data, nullptr, nullptr);
        if (rc != CLLM_E_UNSUPPORTED) return rc;          // very long rows: the two-launch path below
    }

    rc = launch_quantize_act(st, kind, tv(src1), wdata, stride);
    if (rc) return rc;

    const int64_t r2 = src1->ne[2] / src0->ne[2], r3 = src1->ne[3] / src0->ne[3];
    const tview x = tv(src1);
    for (int64_t i13 = 0; i13 < src1->ne[3]; i13++)
    for (int64_t i12 = 0; i12 < src1->ne[2]; i12++) {
        tview w = tv(src0);
        w.data += (i12 / r2) * w.nb[2] + (i13 / r3) * w.nb[3];
        w.ne[2] = w.ne[3] = 1;
        tview d = tv(dst);
        d.data += i12 * d.nb[2] + i13 * d.nb[3];
        const char * act = (const char *) wdata + (size_t)(i12 * src1->ne[1] + i13 * src1->ne[1] * src1->ne[2]) * stride;
        const int path = gemm_path(src1->ne[1], src0->ne[1]);
        if (path) {
            rc = CLLM_E_UNSUPPORTED;
            if (path == 2 && prefill_f16_enabled()) {            // opt-in: dequantize -> dense fp16 GEMM (dense_f16.hip); a different computation than the reference's
                tview x1 = x; x1.data += i12 * x.nb[2] + i13 * x.nb[3];
                rc = launch_dense_f16(st, src0->type, w, x1, d);
            }
            if (rc == CLLM_E_UNSUPPORTED && path == 1) rc = launch_mmx(st, src0->type, w, act, stride, x, d);      // the reference's order on the matrix cores
            if (rc == CLLM_E_UNSUPPORTED && path == 2) rc = launch_mmq(st, src0->type, w, act, stride, x, d);
            if (rc == CLLM_E_UNSUPPORTED) rc = launch_mmvq(st, src0->type, w, act, stride, src1->ne[1], x, d);
        } else {
            rc = launch_mmvq(st, src0->type, w, act, stride, src1->ne[1], x, d);
        }
        if (rc) return rc;
    }
    return CLLM_OK;
}

int mmq_min_cols_get() { return mmq_min_cols(); }
// the number of src1 columns from which cllm_op_mul_mat_ex takes a mat-mul (below it: CLLM_E_UNSUPPORTED); INT_MAX-like when the opt-in fp16 prefill path is on
extern "C" int cllm_mul_mat_ex_min_cols(void) { return prefill_f16_enabled() ? (1 << 30) : exact_gemm_min_cols(0); }

extern "C" int cllm_op_mul_mat_ex(void * stream, const cllm_tensor * src0, const cllm_tensor * src1, cllm_tensor * dst, void * wdata, size_t wsize, int pro,
                                  const cllm_tensor * norm_w, float eps, int epi, const cllm_tensor * resid) {
    if (!src0 || !src1 || !dst) FAIL(CLLM_E_INVALID, "mul_mat_ex: null tensor");
    if ((pro != 0 && pro != 1 && pro != 3 && pro != 4 && pro != 5) || (epi != 0 && epi != 1) || ((pro == 1 || pro == 4) && !norm_w) || (epi && resid)) FAIL(CLLM_E_INVALID, "mul_mat_ex: pro %d epi %d", pro, epi);
    if (!is_quant_type(src0->type) || src0->ne[2] != 1 || src0->ne[3] != 1 || src1->ne[2] != 1 || src1->ne[3] != 1 || src1->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32)
        FAIL(CLLM_E_UNSUPPORTED, "mul_mat_ex: 2-D quantized src0, F32 src1 / dst");
    const int64_t K = src0->ne[0], M = src1->ne[1], Nd = epi ? src0->ne[1] / 2 : src0->ne[1];
    if (src1->ne[0] != (pro == 3 ? 2 * K : K) || (epi && src0->ne[1] % 2) || dst->ne[0] != Nd || dst->ne[1] != M || src1->nb[0] != 4 || dst->nb[0] != 4 || dst->nb[1] % 4) FAIL(CLLM_E_INVALID, "mul_mat_ex: shapes");
    if (resid && (resid->type != CLLM_TYPE_F32 || resid->ne[0] != dst->ne[0] || resid->ne[1] != M || resid->nb[0] != 4 || resid->nb[1] % 4)) FAIL(CLLM_E_INVALID, "mul_mat_ex: resid");
    if (pro == 1 && (norm_w->type != CLLM_TYPE_F32 || norm_w->nb[0] != 4 || norm_w->ne[0] != K || t_nelements(norm_w) != K)) FAIL(CLLM_E_INVALID, "mul_mat_ex: norm weight");
    if (pro == 4 && (norm_w->type != CLLM_TYPE_F32 || norm_w->nb[0] != 4 || norm_w->ne[0] != K || norm_w->ne[1] != M || norm_w->ne[2] != 1 || norm_w->ne[3] != 1)) FAIL(CLLM_E_INVALID, "mul_mat_ex: up tensor");
    if (M < exact_gemm_min_cols(0) || prefill_f16_enabled()) return CLLM_E_UNSUPPORTED;
    if (src0->nb[0] != cllm_type_size(src0->type) || K % cllm_blck_size(src0->type)) FAIL(CLLM_E_INVALID, "mul_mat_ex: src0 rows");
    const int kind = act_kind(src0->type);
    const size_t stride = act_row_bytes(K, kind), need = stride * (size_t) M;
    if (!wdata || wsize < need) FAIL(CLLM_E_INVALID, "mul_mat_ex: wdata too small (%zu < %zu)", wsize, need);
    if ((uintptr_t) wdata % 16 || (uintptr_t) src0->data % 2 || (pro != 5 && ((uintptr_t) src1->data % 16 || src1->nb[1] % 16))) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_ex: operand alignment");
    if (src0->type == CLLM_TYPE_Q4_K && ((uintptr_t) src0->data % 16 || src0->nb[1] % 16)) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_ex: Q4_K rows must be 16-byte aligned");
    if (src0->type == CLLM_TYPE_Q4_1 && ((uintptr_t) src0->data % 4 || src0->nb[1] % 4)) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_ex: Q4_1 rows must be 4-byte aligned");
    hipStream_t st = (hipStream_t) stream;
    int rc = pro == 3 ? launch_quantize_act_silu(st, kind, tv(src1), wdata, stride)
           : pro == 1 ? launch_quantize_act_norm(st, kind, tv(src1), (const float *) norm_w->data, eps, wdata, stride)
           : pro == 4 ? launch_quantize_act_silu2(st, kind, tv(src1), tv(norm_w), wdata, stride)
           : pro == 5 ? CLLM_OK                                      // wdata holds this src1's act rows already (the previous call of this stream quantized them)
           : launch_quantize_act(st, kind, tv(src1), wdata, stride);
    if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "mul_mat_ex: this prologue takes 16-byte aligned rows (norm: of at most 16384 values)");
    if (rc) return rc;
    tview x = tv(src1); if (pro == 3) x.ne[0] = K;
    rc = gemm_path(M, src0->ne[1]) != 2 ? launch_mmx(st, src0->type, tv(src0), wdata, stride, x, tv(dst), resid ? (const float *) resid->data : nullptr, resid ? (int64_t)(resid->nb[1] / 4) : 0, epi)
                             : launch_mmq(st, src0->type, tv(src0), wdata, stride, x, tv(dst), resid ? (const float *) resid->data : nullptr, resid ? (int64_t)(resid->nb[1] / 4) : 0, epi);
    if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "mul_mat_ex: the matrix-core kernel does not take this shape");
    return rc;
}

extern "C" int cllm_bench_mul_mat_kernel(void * stream, const cllm_tensor * src0, void * const * src0_datas, int n_src0, const cllm_tensor * src1,
                                         cllm_tensor * dst, void * wdata, size_t wsize, int iters, float * avg_us) {
    int rc = check_mm(src0, src1, dst, "bench_mul_mat_kernel");
    if (rc) return rc;
    if (!is_quant_type(src0->type) || src0->ne[2] != 1 || src0->ne[3] != 1 || src1->ne[2] != 1 || src1->ne[3] != 1 || !src0_datas || n_src0 <= 0 || iters <= 0 || !avg_us)
        FAIL(CLLM_E_INVALID, "bench_mul_mat_kernel: 2-D quantized operands only");
    hipStream_t st = (hipStream_t) stream;
    const int kind = act_kind(src0->type);
    const size_t stride = act_row_bytes(src0->ne[0], kind);
    if (!wdata || wsize < stride * (size_t) src1->ne[1]) FAIL(CLLM_E_INVALID, "bench_mul_mat_kernel: wdata too small");
    rc = launch_quantize_act(st, kind, tv(src1), wdata, stride);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const int path = gemm_path(src1->ne[1], src0->ne[1]);
    for (int pass = 0; pass < 2; pass++) {             // pass 0 = warm-up (also pages in code objects), pass 1 = timed
        if (pass == 1) HIP_TRY(hipEventRecord(e0, st));
        const int n = pass == 0 ? (n_src0 < 4 ? n_src0 : 4) : iters;
        for (int i = 0; i < n; i++) {
            tview w = tv(src0); w.data = (char *) src0_datas[i % n_src0];
            rc = !path ? CLLM_E_UNSUPPORTED : (path == 2 && prefill_f16_enabled()) ? launch_dense_f16(st, src0->type, w, tv(src1), tv(dst))
                                            : path == 1 ? launch_mmx(st, src0->type, w, wdata, stride, tv(src1), tv(dst)) : launch_mmq(st, src0->type, w, wdata, stride, tv(src1), tv(dst));
            if (rc == CLLM_E_UNSUPPORTED) rc = launch_mmvq(st, src0->type, w, wdata, stride, src1->ne[1], tv(src1), tv(dst));
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    *avg_us = ms * 1e3f / (float) iters;
    return CLLM_OK;
}

// One launch for a node pattern around a single-column quantized MUL_MAT (what a ggml backend's graph_compute can fuse):
//   pro 1: dst = W . quantize(RMS_NORM(px, eps) * pw)        RMS_NORM -> MUL -> MUL_MAT
//   pro 2: dst = W . quantize(px)                             MUL_MAT
//   pro 4: dst = W . quantize(silu(px) * pw)                  UNARY(SILU) -> MUL -> MUL_MAT
//   + resid (may be NULL): dst += resid                       ... -> ADD        (dst may alias resid, not px / pw)
extern "C" int cllm_op_mul_mat_vec_fused(void * stream, const cllm_tensor * src0, int pro, const float * px, const float * pw, float eps, int epi,
                                         const float * resid, float * dst) {
    if (!src0 || !px || !dst || (pro != 1 && pro != 2 && pro != 4) || ((pro == 1 || pro == 4) && !pw) || (epi != 0 && epi != 1)) FAIL(CLLM_E_INVALID, "mul_mat_vec_fused: arguments");
    if (!is_quant_type(src0->type) || src0->ne[2] != 1 || src0->ne[3] != 1 || src0->nb[1] != cllm_row_size(src0->type, src0->ne[0])) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_fused: src0 must be a dense 2-D quantized matrix");
    if (((uintptr_t) px | (uintptr_t) pw | (uintptr_t) src0->data) & 15) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_vec_fused: alignment");
    return launch_gemv_decode((hipStream_t) stream, src0->type, src0->data, src0->ne[0], src0->ne[1], pro, px, pw, eps, epi, dst, nullptr, resid);
}

// The router of a sparse-MoE block for one token (GenericSparseMLP::forward src/layers.cpp:3792-3830: RMS_NORM -> MUL -> MUL_MAT(gate) -> SOFT_MAX -> TOP_K)
// as one launch: xnorm F32 [K] = RMS_NORM(x, eps) * norm_w; probs F32 [n_expert] = SOFT_MAX(gate_w . quantize(xnorm)); ids I32 [k] = TOP_K(probs).
// xnorm may be x itself (in place) but must not overlap it otherwise.  CLLM_E_UNSUPPORTED (nothing launched): shapes for the node-by-node ops.
extern "C" int cllm_op_moe_router(void * stream, const cllm_tensor * x, const cllm_tensor * norm_w, float eps, const cllm_tensor * gate_w,
                                  cllm_tensor * xnorm, cllm_tensor * probs, cllm_tensor * ids) {
    if (!x || !norm_w || !gate_w || !xnorm || !probs || !ids) FAIL(CLLM_E_INVALID, "moe_router: null");
    const int64_t K = gate_w->ne[0], n = gate_w->ne[1], k = ids->ne[0];
    if (x->type != CLLM_TYPE_F32 || norm_w->type != CLLM_TYPE_F32 || xnorm->type != CLLM_TYPE_F32 || probs->type != CLLM_TYPE_F32 || ids->type != CLLM_TYPE_I32 ||
        x->nb[0] != 4 || norm_w->nb[0] != 4 || xnorm->nb[0] != 4 || probs->nb[0] != 4 || ids->nb[0] != 4)
        FAIL(CLLM_E_UNSUPPORTED, "moe_router: dense F32 vectors, I32 ids");
    if (x->ne[0] != K || norm_w->ne[0] != K || xnorm->ne[0] != K || probs->ne[0] != n || t_nelements(x) != K || t_nelements(xnorm) != K || t_nelements(probs) != n || t_nelements(ids) != k)
        FAIL(CLLM_E_INVALID, "moe_router: shapes (one token)");
    if (!is_quant_type(gate_w->type) || gate_w->ne[2] != 1 || gate_w->ne[3] != 1 || gate_w->nb[1] != cllm_row_size(gate_w->type, K)) return CLLM_E_UNSUPPORTED;
    if (((uintptr_t) x->data | (uintptr_t) norm_w->data | (uintptr_t) gate_w->data | (uintptr_t) xnorm->data) & 15) return CLLM_E_UNSUPPORTED;
    if (xnorm->data != x->data && (const char *) xnorm->data < (const char *) x->data + K * 4 && (const char *) x->data < (const char *) xnorm->data + K * 4) FAIL(CLLM_E_INVALID, "moe_router: xnorm overlaps x");
    return launch_moe_router((hipStream_t) stream, gate_w->type, gate_w->data, K, n, (const float *) x->data, (const float *) norm_w->data, eps,
                             (float *) xnorm->data, (float *) probs->data, (int32_t *) ids->data, (int) k);
}

// device-side repack of weight rows for the merged launches: concatenation of n matrices, or the rows of two equally sized ones alternating
extern "C" int cllm_pack_rows(void * stream, void * dst, const void * const * srcs, const int64_t * nrows, int n, size_t row_bytes, int interleave) {
    if (!dst || !srcs || !nrows || n <= 0 || !row_bytes || (interleave && (n != 2 || nrows[0] != nrows[1]))) FAIL(CLLM_E_INVALID, "pack_rows: arguments");
    hipStream_t st = (hipStream_t) stream;
    if (interleave) {
        for (int i = 0; i < 2; i++)
            HIP_TRY(hipMemcpy2DAsync((char *) dst + i * row_bytes, 2 * row_bytes, srcs[i], row_bytes, row_bytes, (size_t) nrows[i], hipMemcpyDeviceToDevice, st));
        return CLLM_OK;
    }
    size_t off = 0;
    for (int i = 0; i < n; i++) {
        if (!srcs[i] || nrows[i] < 0) FAIL(CLLM_E_INVALID, "pack_rows: source %d", i);
        const size_t b = (size_t) nrows[i] * row_bytes;
        if (b) HIP_TRY(hipMemcpyAsync((char *) dst + off, srcs[i], b, hipMemcpyDeviceToDevice, st));
        off += b;
    }
    return CLLM_OK;
}

// times `iters` MUL_MAT_ID launches (quantize of b included, as the op does it), cycling the ids through ids_list[0..n_ids)
extern "C" int cllm_bench_mul_mat_id(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids, void * const * ids_datas, int n_ids,
                                     cllm_tensor * dst, void * wdata, size_t wsize, int iters, float * avg_us) {
    if (!as || !b || !ids || !dst || !ids_datas || n_ids <= 0 || iters <= 0 || !avg_us) FAIL(CLLM_E_INVALID, "bench_mul_mat_id: arguments");
    hipStream_t st = (hipStream_t) stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) HIP_TRY(hipEventRecord(e0, st));
        const int n = pass == 0 ? 2 : iters;
        for (int i = 0; i < n; i++) {
            cllm_tensor id = *ids; id.data = ids_datas[i % n_ids];
            const int rc = cllm_op_mul_mat_id(stream, as, b, &id, dst, wdata, wsize);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    *avg_us = ms * 1e3f / (float) iters;
    return CLLM_OK;
}

// times the DECODE form of the mat-vec (activation prologue inside the kernel, exactly what cllm_llama's fused step launches)
extern "C" int cllm_bench_gemv_fused(void * stream, int wtype, void * const * w_datas, int n_w, int64_t K, int64_t nrows, int pro,
                                     const float * px, const float * pw, float eps, int epi, float * dst, const float * resid, int iters, float * avg_us) {
    if (!is_quant_type(wtype) || !w_datas || n_w <= 0 || iters <= 0 || !avg_us || !px || !dst || pro < 1 || pro > 3) FAIL(CLLM_E_INVALID, "bench_gemv_fused: arguments");
    hipStream_t st = (hipStream_t) stream;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) HIP_TRY(hipEventRecord(e0, st));
        const int n = pass == 0 ? (n_w < 4 ? n_w : 4) : iters;
        for (int i = 0; i < n; i++) {
            const int rc = launch_mmvq_fused(st, wtype, w_datas[i % n_w], K, nrows, pro, px, pw, eps, epi, dst, nullptr, resid);
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    *avg_us = ms * 1e3f / (float) iters;
    return CLLM_OK;
}

// measurement helper (bench.py `ceilings`): what a pure streaming read reaches on this device -- every lane sums 16-byte loads of a buffer larger than the Infinity Cache
// ("record an achieved-copy ceiling" next to the 8 TB/s nominal peak).  4 loads in flight per lane, two 1024-thread workgroups per CU.
__global__ void __launch_bounds__(1024) k_bench_read(const u32x4 * __restrict__ p, size_t n16, unsigned * out) {
    unsigned acc = 0;
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = (size_t) blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += (a.x ^ a.y ^ a.z ^ a.w) + (b.x ^ b.y ^ b.z ^ b.w) + (c.x ^ c.y ^ c.z ^ c.w) + (d.x ^ d.y ^ d.z ^ d.w);
    }
    for (; i < n16; i += stride) { const u32x4 a = p[i]; acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) out[0] = acc;          // (keeps the loads alive; the buffer is filled with 0x01 bytes: never true)
}
extern "C" int cllm_bench_read_bw(void * stream, size_t bytes, int iters, float * gb_per_s) {
    if (!gb_per_s || iters <= 0 || bytes < (1u << 20)) FAIL(CLLM_E_INVALID, "bench_read_bw: arguments");
    hipStream_t st = (hipStream_t) stream;
    char * buf = nullptr; unsigned * out = nullptr;
    HIP_TRY(hipMalloc((void **) &buf, bytes + 16));
    out = (unsigned *)(buf + (bytes & ~(size_t) 15));
    HIP_TRY(hipMemsetAsync(buf, 1, bytes + 16, st));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    const unsigned grid = (unsigned) device_cu_count() * 2;
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) HIP_TRY(hipEventRecord(e0, st));
        for (int i = 0; i < (pass == 0 ? 1 : iters); i++) hipLaunchKernelGGL(k_bench_read, dim3(grid), dim3(1024), 0, st, (const u32x4 *) buf, bytes / 16, out);
    }
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    HIP_TRY(hipFree(buf));
    *gb_per_s = (float)((double)(bytes / 16 * 16) * iters / (ms * 1e-3) / 1e9);
    return CLLM_OK;
}

// gate and up expert mat-vecs of ONE token + UNARY(SILU) + MUL (MultiMLP::forward, src/layers.cpp:3674-3688) in one launch:
// as_gu = the gate and up expert tensors with their rows alternating inside every expert (cllm_pack_rows over [K, F * E], interleave 1):
// [K, 2F, E]; dst[u, slot] = silu(gate_e[u] . x) * (up_e[u] . x), e = ids[slot]; b: [K, 1 | n_used, 1], dst: [F, n_used, 1]
extern "C" int cllm_op_mul_mat_id_silu_mul(void * stream, const cllm_tensor * as_gu, const cllm_tensor * b, const cllm_tensor * ids, cllm_tensor * dst) {
    if (!as_gu || !b || !ids || !dst) FAIL(CLLM_E_INVALID, "mul_mat_id_silu_mul: null");
    if (!is_quant_type(as_gu->type) || b->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || ids->type != CLLM_TYPE_I32) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_id_silu_mul: types");
    const int64_t n_used = ids->ne[0];
    if (ids->ne[1] != 1 || as_gu->ne[0] != b->ne[0] || as_gu->ne[1] % 2 || dst->ne[0] != as_gu->ne[1] / 2 || dst->ne[1] != n_used || dst->ne[2] != 1 || b->ne[2] != 1 ||
        (b->ne[1] != 1 && b->ne[1] != n_used)) FAIL(CLLM_E_INVALID, "mul_mat_id_silu_mul: shapes (one token)");
    if (ids->nb[0] != 4 || b->nb[0] != 4 || dst->nb[0] != 4 || as_gu->nb[1] != cllm_row_size(as_gu->type, as_gu->ne[0]) || b->nb[1] % 16 || dst->nb[1] % 4 ||
        ((uintptr_t) b->data & 15) || ((uintptr_t) as_gu->data & 15) || as_gu->nb[2] % 16) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_id_silu_mul: layout");
    return launch_gemv_decode_id((hipStream_t) stream, as_gu->type, as_gu->data, as_gu->nb[2], as_gu->ne[0], as_gu->ne[1], (const float *) b->data,
                                 b->ne[1] == 1 ? 0 : (int64_t)(b->nb[1] / 4), (const int32_t *) ids->data, (int) n_used, (float *) dst->data, (int64_t)(dst->nb[1] / 4), 1);
}

// The router of a sparse-MoE block AND the experts' gate / up projections of ONE token in one launch (gemv_moe.hip, EPI 5): cllm_op_moe_router + cllm_op_mul_mat_id_silu_mul
// without the router launch and without the normalised activation in memory.  x [K] F32 (the block's input), norm_w [K] F32, gate_w [K, E] quantized (the router),
// as_gu [K, 2F, E] the per-expert interleaved gate / up pack of the SAME type, probs [E] F32 and ids [k] I32 written (for cllm_op_mul_mat_id_combine), dst [F, k] F32.
// Bit-identical to the two calls.  CLLM_E_UNSUPPORTED (nothing launched): make the two calls.
extern "C" int cllm_op_moe_router_gate_up(void * stream, const cllm_tensor * x, const cllm_tensor * norm_w, float eps, const cllm_tensor * gate_w, const cllm_tensor * as_gu,
                                          cllm_tensor * probs, cllm_tensor * ids, cllm_tensor * dst) {
    if (!x || !norm_w || !gate_w || !as_gu || !probs || !ids || !dst) FAIL(CLLM_E_INVALID, "moe_router_gate_up: null");
    const int64_t K = gate_w->ne[0], E = gate_w->ne[1], k = ids->ne[0];
    if (x->type != CLLM_TYPE_F32 || norm_w->type != CLLM_TYPE_F32 || probs->type != CLLM_TYPE_F32 || ids->type != CLLM_TYPE_I32 || dst->type != CLLM_TYPE_F32 ||
        x->nb[0] != 4 || norm_w->nb[0] != 4 || probs->nb[0] != 4 || ids->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_UNSUPPORTED, "moe_router_gate_up: dense F32 vectors, I32 ids");
    if (x->ne[0] != K || t_nelements(x) != K || norm_w->ne[0] != K || probs->ne[0] != E || t_nelements(probs) != E || t_nelements(ids) != k || as_gu->ne[0] != K || as_gu->ne[2] != E ||
        as_gu->ne[1] % 2 || dst->ne[0] != as_gu->ne[1] / 2 || dst->ne[1] != k || dst->ne[2] != 1) FAIL(CLLM_E_INVALID, "moe_router_gate_up: shapes (one token)");
    if (!is_quant_type(gate_w->type) || gate_w->type != as_gu->type || gate_w->ne[2] != 1 || gate_w->ne[3] != 1 || gate_w->nb[1] != cllm_row_size(gate_w->type, K) ||
        as_gu->nb[1] != cllm_row_size(as_gu->type, K) || as_gu->nb[2] % 16 || dst->nb[1] % 4) return CLLM_E_UNSUPPORTED;
    if (((uintptr_t) x->data | (uintptr_t) norm_w->data | (uintptr_t) gate_w->data | (uintptr_t) as_gu->data) & 15) return CLLM_E_UNSUPPORTED;
    return launch_gemv_decode_id_router_silu((hipStream_t) stream, as_gu->type, as_gu->data, as_gu->nb[2], K, as_gu->ne[1], (const float *) x->data, (const float *) norm_w->data, eps,
                                             gate_w->data, (int) E, (int) k, (float *) probs->data, (int32_t *) ids->data, (float *) dst->data, (int64_t)(dst->nb[1] / 4));
}

// MUL_MAT_ID(down experts) of ONE token over TWO slots + the tail of the sparse-MoE block (GenericSparseMLP::forward src/layers.cpp:3840-3872) in one launch:
//   dst[r] = (as[ids[0]][r] . b[:, 0]) * w0 + (as[ids[1]][r] . b[:, 1]) * w1 (+ resid[r]),   w_j = probs[ids[j]] / (probs[ids[0]] + probs[ids[1]])
// as [K, H, E] quantized, b F32 [K, 2, 1], ids I32 [2, 1], probs F32 [E, 1], resid / dst F32 [H, 1]; dst may be resid itself, nothing else.
// Bit-identical to cllm_op_mul_mat_id -> cllm_op_moe_combine.  CLLM_E_UNSUPPORTED (nothing launched): use those two.
extern "C" int cllm_op_mul_mat_id_combine(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids, const cllm_tensor * probs,
                                          const cllm_tensor * resid, cllm_tensor * dst) {
    if (!as || !b || !ids || !probs || !dst) FAIL(CLLM_E_INVALID, "mul_mat_id_combine: null");
    if (!is_quant_type(as->type) || b->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || ids->type != CLLM_TYPE_I32 || probs->type != CLLM_TYPE_F32 || (resid && resid->type != CLLM_TYPE_F32))
        FAIL(CLLM_E_UNSUPPORTED, "mul_mat_id_combine: types");
    const int64_t K = as->ne[0], H = as->ne[1], E = as->ne[2];
    if (b->ne[0] != K || b->ne[2] != 1 || b->ne[3] != 1 || ids->ne[1] != 1 || t_nelements(ids) != ids->ne[0] || probs->ne[0] != E || t_nelements(probs) != E || dst->ne[0] != H || t_nelements(dst) != H ||
        (resid && (resid->ne[0] != H || t_nelements(resid) != H))) FAIL(CLLM_E_INVALID, "mul_mat_id_combine: shapes (one token)");
    if (ids->ne[0] != 2 || b->ne[1] != 2) return CLLM_E_UNSUPPORTED;
    if (ids->nb[0] != 4 || b->nb[0] != 4 || dst->nb[0] != 4 || probs->nb[0] != 4 || (resid && resid->nb[0] != 4) || as->nb[1] != cllm_row_size(as->type, K) || b->nb[1] % 16 ||
        ((uintptr_t) b->data & 15) || ((uintptr_t) as->data & 15) || as->nb[2] % 16) return CLLM_E_UNSUPPORTED;
    return launch_gemv_decode_id_combine((hipStream_t) stream, as->type, as->data, as->nb[2], K, H, (const float *) b->data, (int64_t)(b->nb[1] / 4), (const int32_t *) ids->data,
                                         (const float *) probs->data, resid ? (const float *) resid->data : nullptr, (float *) dst->data);
}

extern "C" int cllm_op_mul_mat_id(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids, cllm_tensor * dst,
                                  void * wdata, size_t wsize) {
    int rc = check_mm(as, b, dst, "mul_mat_id");
    if (rc) return rc;
    if (!ids || ids->type != CLLM_TYPE_I32) FAIL(CLLM_E_INVALID, "mul_mat_id: ids must be I32");
    if (!is_quant(as->type)) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_id: expert weights must be a quantized type");
    const int64_t n_used = ids->ne[0], n_tok = ids->ne[1];
    if (as->ne[3] != 1 || b->ne[3] != 1 || dst->ne[3] != 1 || ids->ne[2] != 1 || ids->ne[3] != 1) FAIL(CLLM_E_INVALID, "mul_mat_id: 4-D operands");
    if (dst->ne[0] != as->ne[1] || dst->ne[1] != n_used || dst->ne[2] != n_tok || b->ne[2] != n_tok) FAIL(CLLM_E_INVALID, "mul_mat_id: shapes");
    if (b->ne[1] <= 0 || (b->ne[1] != 1 && b->ne[1] != n_used)) FAIL(CLLM_E_INVALID, "mul_mat_id: b.ne[1] must be 1 or n_expert_used");
    if (t_nelements(dst) == 0) return CLLM_OK;
    const int kind = act_kind(as->type);
    const size_t stride = act_row_bytes(as->ne[0], kind);
    const size_t need = stride * (size_t) t_nrows(b);
    if (!wdata || wsize < need) FAIL(CLLM_E_INVALID, "mul_mat_id: wdata too small (%zu < %zu)", wsize, need);
    if (as->type == CLLM_TYPE_Q4_K && ((uintptr_t) as->data % 16 || as->nb[1] % 16 || as->nb[2] % 16)) FAIL(CLLM_E_UNSUPPORTED, "mul_mat_id: alignment");
    hipStream_t st = (hipStream_t) stream;
    // one token (decode): the decode mat-vec with the expert picked per grid slice, activation quantized inside (one launch per MUL_MAT_ID)
    if (n_tok == 1 && ids->nb[0] == 4 && b->nb[0] == 4 && dst->nb[0] == 4 && as->nb[1] == cllm_row_size(as->type, as->ne[0]) && b->nb[1] % 16 == 0 && dst->nb[1] % 4 == 0 &&
        ((uintptr_t) b->data & 15) == 0 && ((uintptr_t) as->data & 15) == 0 && as->nb[2] % 16 == 0) {
        rc = launch_gemv_decode_id(st, as->type, as->data, as->nb[2], as->ne[0], as->ne[1], (const float *) b->data, b->ne[1] == 1 ? 0 : (int64_t)(b->nb[1] / 4),
                                   (const int32_t *) ids->data, (int) n_used, (float *) dst->data, (int64_t)(dst->nb[1] / 4));
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    rc = launch_quantize_act(st, kind, tv(b), wdata, stride);      // act row index = i11 + ne11*i12 (token-major over slots)
    if (rc) return rc;
    if (is_kq_type(as->type)) {                                     // the coverage types: one grid slice per (token, slot) of gemv_kq.hip
        rc = launch_gemv_kq_id(st, as->type, tv(as), wdata, stride, b->ne[1], tv(ids), tv(dst));
        if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "mul_mat_id: type %d: shape or alignment not taken", as->type);
        return rc;
    }
    return launch_mmvq_id(st, as->type, tv(as), wdata, stride, b->ne[1], tv(ids), tv(dst));
}

// ================================================================================================
// flash attention (fattn.hip)
// ================================================================================================
extern "C" size_t cllm_flash_attn_wsize(const cllm_tensor * q) {
    return q ? fattn_wsize(q->ne[1], q->ne[2], q->ne[3], q->ne[0]) : 0;
}
extern "C" int cllm_op_flash_attn_ext(void * stream, const cllm_tensor * q, const cllm_tensor * k, const cllm_tensor * v, const cllm_tensor * mask,
                                      cllm_tensor * dst, float scale, float max_bias, float logit_softcap, void * wdata, size_t wsize) {
    if (!q || !k || !v || !dst || !q->data || !k->data || !v->data || !dst->data) FAIL(CLLM_E_INVALID, "flash_attn_ext: null tensor");
    if (max_bias != 0.0f || logit_softcap != 0.0f) FAIL(CLLM_E_UNSUPPORTED, "flash_attn_ext: ALiBi / logit soft-cap are not on this path");
    if (q->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || k->type != v->type || (mask && mask->type != CLLM_TYPE_F16)) FAIL(CLLM_E_UNSUPPORTED, "flash_attn_ext: types");
    if (dst->ne[0] != v->ne[0] || dst->ne[1] != q->ne[2] || dst->ne[2] != q->ne[1] || dst->ne[3] != q->ne[3]) FAIL(CLLM_E_INVALID, "flash_attn_ext: dst shape");
    if (dst->nb[0] != 4 || k->nb[0] != cllm_type_size(k->type) || v->nb[0] != cllm_type_size(v->type)) FAIL(CLLM_E_UNSUPPORTED, "flash_attn_ext: rows must be dense");
    if (mask && (mask->ne[0] < k->ne[1] || mask->ne[1] < q->ne[1] || (mask->ne[2] != 1 && mask->ne[2] != q->ne[2]) || (mask->ne[3] != 1 && mask->ne[3] != q->ne[3])))
        FAIL(CLLM_E_INVALID, "flash_attn_ext: mask shape");
    const tview m = mask ? tv(mask) : tview();
    // dst [D, H, N, B]: element (dv, h, n, b) at n nb[2] + h nb[1] + b nb[3]
    const int rc = launch_fattn((hipStream_t) stream, tv(q), tv(k), k->type, tv(v), 0, mask ? &m : nullptr, -1, (char *) dst->data,
                                (int64_t) dst->nb[2], (int64_t) dst->nb[1], (int64_t) dst->nb[3], scale, wdata, wsize);
    if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "flash_attn_ext: shape / layout not taken (D %lld, n_kv %lld)", (long long) q->ne[0], (long long) k->ne[1]);
    return rc;
}
extern "C" int cllm_attn_prefill_min_cols(void) { return flash_prefill_min_cols(); }
extern "C" int cllm_op_attn_prefill(void * stream, const cllm_tensor * q, const cllm_tensor * k, const cllm_tensor * vt, cllm_tensor * dst, float scale, int n_past) {
    if (!q || !k || !vt || !dst || !q->data || !k->data || !vt->data || !dst->data) FAIL(CLLM_E_INVALID, "attn_prefill: null tensor");
    if (q->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || k->type != CLLM_TYPE_F16 || vt->type != CLLM_TYPE_F16 || n_past < 0) FAIL(CLLM_E_UNSUPPORTED, "attn_prefill: types");
    if (dst->ne[0] != vt->ne[1] || dst->ne[1] != q->ne[1] || dst->ne[2] != q->ne[2] || dst->ne[3] != q->ne[3] || dst->nb[0] != 4 || k->nb[0] != 2 || vt->nb[0] != 2) FAIL(CLLM_E_INVALID, "attn_prefill: shapes");
    if (k->ne[1] != n_past + q->ne[1]) FAIL(CLLM_E_INVALID, "attn_prefill: n_kv != n_past + qlen");
    if (prefill_attn_mode() == 1 && q->ne[3] == 1 && k->ne[3] == 1) {       // the reference's order: K.Q, soft_max, V.P on the exact kernels (mmf_exact.hip)
        tview ve = tv(vt); ve.ne[0] = k->ne[1];
        const int rc = attn_prefill_exact((hipStream_t) stream, tv(q), tv(k), ve, (char *) dst->data, (int64_t) dst->nb[1], (int64_t) dst->nb[2], scale, n_past);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    tview v = tv(vt);
    v.ne[0] = v.nb[1] / 2;                       // the row really holds max_length positions; n_kv comes from k
    const int rc = launch_fattn((hipStream_t) stream, tv(q), tv(k), k->type, v, 1, nullptr, n_past, (char *) dst->data,
                                (int64_t) dst->nb[1], (int64_t) dst->nb[2], (int64_t) dst->nb[3], scale, nullptr, 0);
    if (rc == CLLM_E_UNSUPPORTED) FAIL(rc, "attn_prefill: shape / layout not taken");
    return rc;
}
