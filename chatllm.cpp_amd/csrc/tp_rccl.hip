// tp_rccl.hip -- tensor-parallel all-reduce of the residual-stream partial sums over RCCL (xGMI), on the runner's own stream
// so that it is part of the captured decode graph (no host round trip per collective).
//
// The reference has no tensor parallelism (SplitMethod::Row is a TODO, src/backend.cpp:677-778 only splits by layer); this is
// q/k/v/gate/up sharded by rows, o/down by columns, one all-reduce(sum) of [hidden] fp32 after o_proj and
// after down_proj.  librccl is opened with dlopen: the single-GPU library has no link-time dependency on it.
#include "common.h"
#include <dlfcn.h>
#include <string.h>

typedef struct { char internal[128]; } tp_unique_id;             // = ncclUniqueId (NCCL_UNIQUE_ID_BYTES 128)
typedef int (*fn_get_unique_id)(tp_unique_id *);
typedef int (*fn_comm_init_rank)(void **, int, tp_unique_id, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_all_reduce)(const void *, void *, size_t, int, int, void *, hipStream_t);
typedef const char * (*fn_get_error_string)(int);
typedef int (*fn_comm_count)(const void *, int *);

static struct {
    void * so; fn_get_unique_id get_unique_id; fn_comm_init_rank comm_init_rank; fn_comm_destroy comm_destroy; fn_all_reduce all_reduce;
    fn_get_error_string get_error_string; fn_comm_count comm_count, comm_user_rank;
} g_rccl;

static int rccl_load() {
    if (g_rccl.so) return CLLM_OK;
    void * so = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!so) so = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!so) FAIL(CLLM_E_UNSUPPORTED, "tp: cannot open librccl.so (%s)", dlerror());
    g_rccl.get_unique_id    = (fn_get_unique_id)    dlsym(so, "ncclGetUniqueId");
    g_rccl.comm_init_rank   = (fn_comm_init_rank)   dlsym(so, "ncclCommInitRank");
    g_rccl.comm_destroy     = (fn_comm_destroy)     dlsym(so, "ncclCommDestroy");
    g_rccl.all_reduce       = (fn_all_reduce)       dlsym(so, "ncclAllReduce");
    g_rccl.get_error_string = (fn_get_error_string) dlsym(so, "ncclGetErrorString");
    g_rccl.comm_count       = (fn_comm_count)       dlsym(so, "ncclCommCount");
    g_rccl.comm_user_rank   = (fn_comm_count)       dlsym(so, "ncclCommUserRank");
    if (!g_rccl.get_unique_id || !g_rccl.comm_init_rank || !g_rccl.comm_destroy || !g_rccl.all_reduce) { dlclose(so); FAIL(CLLM_E_UNSUPPORTED, "tp: librccl.so lacks the NCCL entry points"); }
    g_rccl.so = so;
    return CLLM_OK;
}
static int rccl_fail(int rc, const char * what) {
    FAIL(CLLM_E_HIP, "tp: %s failed: %s (%d)", what, g_rccl.get_error_string ? g_rccl.get_error_string(rc) : "?", rc);
}

extern "C" int cllm_tp_unique_id(void * out128) {
    if (!out128) FAIL(CLLM_E_INVALID, "tp_unique_id: null");
    { const int rc_ = rccl_load(); if (rc_) return rc_; }
    tp_unique_id id;
    const int rc = g_rccl.get_unique_id(&id);
    if (rc) return rccl_fail(rc, "ncclGetUniqueId");
    memcpy(out128, &id, sizeof id);
    return CLLM_OK;
}
extern "C" int cllm_tp_init(const void * id128, int rank, int nranks, void ** comm_out) {
    if (!id128 || !comm_out || nranks < 1 || rank < 0 || rank >= nranks) FAIL(CLLM_E_INVALID, "tp_init: arguments");
    { const int rc_ = rccl_load(); if (rc_) return rc_; }
    tp_unique_id id; memcpy(&id, id128, sizeof id);
    void * comm = nullptr;
    const int rc = g_rccl.comm_init_rank(&comm, nranks, id, rank);       // collective: every rank of the group calls it
    if (rc) return rccl_fail(rc, "ncclCommInitRank");
    *comm_out = comm;
    return CLLM_OK;
}
// what the communicator itself reports: ranks in the group and this rank's index (bench.py prints them: the run's parallelism as RCCL sees it, not as the launcher meant it)
extern "C" int cllm_tp_comm_info(void * comm, int * nranks, int * rank) {
    if (!comm || !nranks || !rank) FAIL(CLLM_E_INVALID, "tp_comm_info: null");
    { const int rc_ = rccl_load(); if (rc_) return rc_; }
    if (!g_rccl.comm_count || !g_rccl.comm_user_rank) FAIL(CLLM_E_UNSUPPORTED, "tp_comm_info: librccl.so lacks ncclCommCount / ncclCommUserRank");
    int rc = g_rccl.comm_count(comm, nranks);
    if (rc) return rccl_fail(rc, "ncclCommCount");
    rc = g_rccl.comm_user_rank(comm, rank);
    if (rc) return rccl_fail(rc, "ncclCommUserRank");
    return CLLM_OK;
}
extern "C" int cllm_tp_destroy(void * comm) {
    if (!comm) return CLLM_OK;
    { const int rc_ = rccl_load(); if (rc_) return rc_; }
    const int rc = g_rccl.comm_destroy(comm);
    if (rc) return rccl_fail(rc, "ncclCommDestroy");
    return CLLM_OK;
}
// in-place sum of n floats over the group, stream-ordered on `stream` (capturable in a hipGraph)
extern "C" int cllm_tp_all_reduce_f32(void * comm, void * stream, float * buf, size_t n) {
    if (!comm || !buf) FAIL(CLLM_E_INVALID, "tp_all_reduce: null");
    const int rc = g_rccl.all_reduce(buf, buf, n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, (hipStream_t) stream);
    if (rc) return rccl_fail(rc, "ncclAllReduce");
    return CLLM_OK;
}
