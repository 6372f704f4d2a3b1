// ggml-hip.cpp -- the reference-side binding: a ggml backend MODULE ("libggml-hip.so") over the C ABI of
// libchatllm_hip.so (include/chatllm_hip.h).  This is the drop-in boundary of SURVEY.md 8b:
//   * exports  ggml_backend_reg_t ggml_backend_init(void)            (ggml/src/ggml-backend-impl.h:206-251)
//   * implements the five vtables reg / device / buffer_type / buffer / backend   (ggml-backend-impl.h:11-251)
//   * is found by the unmodified chatllm.cpp host through ggml_backend_load_all_from_path(), slot "hip", BEFORE the
//     CPU module (ggml/src/ggml-backend-reg.cpp:545-575), so CPU stays the last device (src/backend.cpp:727-737).
// It is compiled against the reference's headers WHERE THEY LIE (/root/reference/ggml/include, ggml/src) -- nothing is
// copied -- and contains no kernels: graph_compute() walks the cgraph and maps every node 1:1 onto a cllm_op_* call.
// Unsupported nodes are declined in supports_op() so that ggml's scheduler places them on the CPU backend.
#include "ggml.h"
#include "ggml-backend.h"
#include "ggml-backend-impl.h"
#include "ggml-impl.h"          // ggml_node_get_use_count: the graph-wide use counts the fusion pass relies on

#include "../../include/chatllm_hip.h"

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <vector>
#include <functional>

#define HIPB_LOG(...) do { fprintf(stderr, "[ggml-hip] " __VA_ARGS__); fputc('\n', stderr); } while (0)

namespace {

struct hip_device_ctx { int id; std::string name, desc; ggml_backend_buffer_type buft; };
struct hip_backend_ctx {
    int device; bool kernel_error = false; void * stream = nullptr; void * wdata = nullptr; size_t wsize = 0; void * abuf = nullptr; size_t asize = 0; void * copy_event = nullptr;
    // replay of a token's launch list (graph_compute): the serialized arguments of every C-ABI call of the last graph, and the
    // captured graph of the list that came twice in a row
    std::vector<uint8_t> last_sig, graph_sig; void * graph_exec = nullptr; bool graph_broken = false; long replays = 0, captures = 0;
    // decode-ahead (see ahead_launch): the next step of a greedy decode started while the host is still busy with the current token
    struct scalar_set { const void * ptr; int32_t val; };
    struct {
        bool armed = false, inflight = false;
        bool chained = false;          // the step running ahead was queued BEHIND the step the host is still waiting for (ahead_launch(c, true)): synchronize() waits for `ev`, not for the stream
        bool tp = false;               // the step armed / running ahead is a tensor-parallel one (replayed through tp_ahead_replay, not graph_exec)
        bool ev_pending = false;       // ... and that event has not been waited for yet (EVERY synchronize() while the step runs ahead must stay off the stream: the host calls it more than once per token)
        bool snap_ready = false;       // the outputs' snapshot is complete although the step did not start (ahead_launch failed behind its argmax write): serve the pending read from it
        void * ev = nullptr, * table_dev = nullptr, * scratch = nullptr, * snap = nullptr; size_t snap_bytes = 0; int32_t * tok_host = nullptr;
        struct snap_rng { const void * src; void * dst; uint64_t bytes; }; void * rng_dev = nullptr; snap_rng * rng_host = nullptr; std::vector<snap_rng> rng_cur;      // the snapshot's ranges as the device holds them
        struct set_rec { const void * ptr; int32_t val, pad; }; set_rec * tab_host = nullptr;      // page-locked: the (pointer, absolute value) records of the step started ahead
        const void * ids_ptr = nullptr, * logits_ptr = nullptr; size_t logits_bytes = 0;
        struct out_range { const char * ptr; size_t bytes, snap_off; }; std::vector<out_range> outs;      // every OUTPUT tensor of the graph (snapshots)
        std::vector<scalar_set> last_sets, pred;
        int misses = 0; long graphs = 0, skip_until = 0, hits = 0, launched = 0;
    } ahead;
};
struct hip_buffer_ctx { int device; void * base; uint64_t uid; std::atomic<uint64_t> gen{0}; };     // gen: bumped by every write through the buffer interface
std::atomic<uint64_t> g_next_buffer_uid{1};     // (accessory models own their own contexts and may run on other threads)

// CLLM_HIP_STATS=1: where a token's wall time goes, seen from the module (host = everything between our entry points: the reference's
// graph build, scheduler split and allocation, sampling)
struct wall_stats {
    using clk = std::chrono::steady_clock;
    clk::time_point last_exit = clk::now();
    double host_us = 0, plan_us = 0, issue_us = 0, sync_us = 0, set_us = 0, get_us = 0, alloc_us = 0; long graphs = 0, calls = 0, sets = 0, gets = 0, allocs = 0, syncs = 0;
    static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
} g_ws;
const bool g_stats = getenv("CLLM_HIP_STATS") != nullptr;
struct ws_scope {        // time inside one of our entry points goes to `slot`, the time since the previous one to host_us
    double & slot; wall_stats::clk::time_point t0;
    explicit ws_scope(double & s) : slot(s), t0(wall_stats::clk::now()) { if (g_stats) g_ws.host_us += wall_stats::us(g_ws.last_exit, t0); }
    ~ws_scope() { if (g_stats) { const auto t1 = wall_stats::clk::now(); slot += wall_stats::us(t0, t1); g_ws.last_exit = t1; } }
};

std::atomic<uint64_t> g_tp_kv_epoch{1};          // tensor parallel (tp_graph_compute): bumped by everything that may have changed the host's KV caches behind the ranks' shards
struct tp_piece { const void * src; size_t spitch, width, rows, doff, dpitch; };
struct tp_shard { void * data = nullptr; size_t bytes = 0; int n = 0; const void * src[3] = {}; uint64_t uid[3] = {}, gen[3] = {}; };
struct tp_rank {
    int gpu = 0; void * stream = nullptr; bool own_stream = false; void * fused = nullptr, * ev = nullptr;
    char * scratch = nullptr; size_t scratch_bytes = 0;
    std::unordered_map<const void *, tp_shard> shards;                   // keyed by the first source tensor's data pointer (a weight tensor has one role)
    std::vector<const void *> kv_sig; std::vector<void *> kv_mem; void * kv_table = nullptr; int64_t kv_valid = 0; uint64_t kv_epoch = 0; size_t kv_dims[3] = {};
};
struct tp_group {
    int n = 0; std::vector<tp_rank> r; bool fused_ready = false, broken = false, told = false, last_tp = false; int sites = 0; size_t max_n = 0; void * ev0 = nullptr;
    long steps = 0, plain = 0, replays = 0, captures = 0; const void * owner = nullptr;
    // what a step started AHEAD of the host needs besides the captured graphs (tp_ahead_replay): the embedding gather's operands, where the position lives, every rank's inputs
    struct { cllm_tensor ea, eb, ed; const void * pos_src = nullptr; size_t hbytes = 0; std::vector<void *> xa, pos; std::vector<void *> streams; std::vector<int> sgpu; int64_t n_kv = 0; bool valid = false; } rp;
    std::vector<uint64_t> last_sig, graph_sig; std::vector<void *> execs; bool graph_broken = false;      // launch-list replay of the sharded step: one captured graph per distinct stream      // owner: the backend context whose stream rank 0 runs on
} g_tp;
std::vector<hip_device_ctx *> g_devices;
ggml_backend_reg g_reg;

cllm_tensor desc(const ggml_tensor * t) {
    cllm_tensor d;
    d.type = (int32_t) t->type;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    d.data = t->data;
    return d;
}
bool is_q(ggml_type t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q4_1 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K; }
// the coverage types: mat-mul and GET_ROWS only (gemv_kq.hip): no fused launch takes them
bool is_kq(ggml_type t) {
    return t == GGML_TYPE_Q5_K || t == GGML_TYPE_Q6_K || t == GGML_TYPE_Q2_K || t == GGML_TYPE_Q3_K || t == GGML_TYPE_Q5_0 || t == GGML_TYPE_Q5_1 || t == GGML_TYPE_IQ4_NL || t == GGML_TYPE_MXFP4 || t == GGML_TYPE_IQ4_XS || t == GGML_TYPE_TQ1_0 || t == GGML_TYPE_TQ2_0 ||
           t == GGML_TYPE_IQ2_XXS || t == GGML_TYPE_IQ2_XS || t == GGML_TYPE_IQ2_S || t == GGML_TYPE_IQ3_XXS || t == GGML_TYPE_IQ3_S || t == GGML_TYPE_IQ1_S || t == GGML_TYPE_IQ1_M;
}
bool dense_rows(const ggml_tensor * t) { return t->nb[0] == ggml_type_size(t->type); }
bool f32_dense(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->nb[0] == 4; }

// ---------------------------------------------------------------------------------------------------------------------------
// buffer
// ---------------------------------------------------------------------------------------------------------------------------
// Small set_tensor calls (chatllm writes the position vector of EVERY layer and the token id before each graph: 33 four-byte writes
// per token for Llama-3-8B, 14 us each as a synchronous copy) go through a page-locked ring: the caller's bytes are consumed when
// buf_set returns, the H2D copies are queued on the device's null stream and waited for ONCE, by whoever touches device memory next
// (graph_compute, get_tensor, cpy_tensor, synchronize, free_buffer).
// What the host last wrote into 4-byte tensors (position vectors, token id), by device address: chatllm gives every layer its own
// position tensor, all holding the same value -- knowing that, graph_compute builds the RoPE cos/sin table once per graph instead of
// once per layer (32 launches of 5 us per token for Llama-3-8B).  Any other write into the range forgets the entry.
std::unordered_map<const void *, int32_t> g_i32_vals;
// the 4-byte I32 tensors the host wrote since the last graph_compute, per device, in order (token id + one position per layer for chatllm):
// what decode-ahead predicts for the next step and checks its prediction against
std::vector<hip_backend_ctx::scalar_set> g_scalar_sets[64];
hip_backend_ctx * g_ahead_ctx[64] = {};          // the backend whose captured step may be running ahead on that device
// a write through the buffer interface that is not one of those scalars must not race a step running ahead (it may target memory that step uses)
void ahead_finish_event(hip_backend_ctx * c);
void ahead_quiesce(int device) {
    hip_backend_ctx * c = device >= 0 && device < 64 ? g_ahead_ctx[device] : nullptr;
    if (c && c->ahead.inflight) { ahead_finish_event(c); cllm_set_device(c->device); cllm_stream_sync(c->stream); }
}
void i32_forget(const void * lo, size_t n) {          // caller holds g_ring.m
    if (g_i32_vals.empty()) return;
    const char * a = (const char *) lo, * b = a + n;
    for (auto it = g_i32_vals.begin(); it != g_i32_vals.end();) { const char * k = (const char *) it->first; if (k + 4 > a && k < b) it = g_i32_vals.erase(it); else ++it; }
}
// Two page-locked rings share the scheme: `small` for the per-token scalars above, `bulk` for the loader -- chatllm uploads a model in 1,024,000-byte
// slices (TensorInfo::read_tensor_data, src/chat.cpp:1322-1338): a blocking copy + synchronize per slice leaves the DMA engine idle while the host
// reads the next slice from the file.  Through the bulk ring the slice is copied into pinned memory (its bytes are consumed when buf_set returns,
// as the interface requires), the H2D copy is queued and the host goes on reading; the ring is waited for only when it wraps (every 64 MiB).
struct set_ring {
    size_t cap, max; char * base = nullptr; size_t head = 0; bool failed = false; bool pending[64] = {}; std::mutex m;
    set_ring(size_t cap_, size_t max_) : cap(cap_), max(max_) {}
};
set_ring g_ring(256u << 10, 4096), g_bulk(64u << 20, 8u << 20);
void flush_ring(set_ring & r) {
    std::lock_guard<std::mutex> lock(r.m);
    for (int d = 0; d < 64; d++) if (r.pending[d]) {
        cllm_set_device(d);
        if (cllm_stream_sync(nullptr) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] queued set_tensor copies on device %d failed: %s\n", d, cllm_last_error());
        r.pending[d] = false;
    }
}
void flush_sets() { flush_ring(g_ring); flush_ring(g_bulk); }
bool ring_set(set_ring & r, int device, void * dst, const void * data, size_t size) {      // device already current
    std::lock_guard<std::mutex> lock(r.m);
    if (!r.base && !r.failed) { void * p = nullptr; if (cllm_host_malloc(&p, r.cap) == CLLM_OK) r.base = (char *) p; else r.failed = true; }
    if (!r.base || device < 0 || device >= 64 || size > r.max) return false;
    const size_t slot = (size + 63) & ~(size_t) 63;
    if (r.head + slot > r.cap) {            // wrap: every queued copy must have read its slot
        for (int d = 0; d < 64; d++) if (r.pending[d]) { cllm_set_device(d); cllm_stream_sync(nullptr); r.pending[d] = false; }
        cllm_set_device(device);
        r.head = 0;
    }
    memcpy(r.base + r.head, data, size);
    if (cllm_memcpy_h2d(dst, r.base + r.head, size, nullptr) != CLLM_OK) return false;
    r.head += slot; r.pending[device] = true;
    return true;
}
// a host buffer of any size into device memory: through the bulk ring (asynchronous), else a blocking copy
bool upload(int device, void * dst, const void * data, size_t size) {
    static const bool sync_loader = getenv("CLLM_HIP_SYNC_LOAD") != nullptr;       // (A/B: the blocking path)
    if (!sync_loader && ring_set(g_bulk, device, dst, data, size)) return true;
    return cllm_memcpy_h2d(dst, data, size, nullptr) == CLLM_OK && cllm_stream_sync(nullptr) == CLLM_OK;
}

void packs_forget(uint64_t uid);
void buf_free(ggml_backend_buffer_t b) {
    ws_scope ws(g_ws.alloc_us); g_ws.allocs++;
    flush_sets();
    auto * c = (hip_buffer_ctx *) b->context;
    ahead_quiesce(c->device);                      // a step running ahead may still use this memory
    if (c->device >= 0 && c->device < 64 && g_ahead_ctx[c->device]) g_ahead_ctx[c->device]->ahead.armed = false;
    { std::lock_guard<std::mutex> lock(g_ring.m); i32_forget(c->base, b->size); }
    cllm_set_device(c->device); packs_forget(c->uid); cllm_free(c->base); delete c;
}
void * buf_base(ggml_backend_buffer_t b) { return ((hip_buffer_ctx *) b->context)->base; }
void buf_memset(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t v, size_t off, size_t size) {
    ahead_quiesce(((hip_buffer_ctx *) b->context)->device);
    cllm_set_device(((hip_buffer_ctx *) b->context)->device); ((hip_buffer_ctx *) b->context)->gen++; g_tp_kv_epoch++;
    { std::lock_guard<std::mutex> lock(g_ring.m); i32_forget((char *) t->data + off, size); }
    if (cllm_memset((char *) t->data + off, v, size, nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] memset_tensor '%s' failed: %s\n", t->name, cllm_last_error());
}
void buf_set(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t off, size_t size) {
    ws_scope ws(g_ws.set_us); g_ws.sets++;
    // called with 1,024,000-byte slices at arbitrary offsets while a model loads (src/chat.cpp:1322-1338): layout stays native
    auto * c = (hip_buffer_ctx *) b->context;
    cllm_set_device(c->device); c->gen++;
    bool scalar = true;
    {
        std::lock_guard<std::mutex> lock(g_ring.m);
        i32_forget((char *) t->data + off, size);
        if (size == 4 && off == 0 && t->type == GGML_TYPE_I32 && g_i32_vals.size() < 4096) memcpy(&g_i32_vals[t->data], data, 4);
        if (size == 4 && off == 0 && t->type == GGML_TYPE_I32 && c->device < 64) {
            hip_backend_ctx::scalar_set ss; ss.ptr = t->data; memcpy(&ss.val, data, 4);
            if (g_scalar_sets[c->device].size() < 1024) g_scalar_sets[c->device].push_back(ss);
            else scalar = false;                          // not recorded: must not be held back either (written through below, after the step running ahead has finished)
        } else scalar = false;
    }
    if (!scalar) { ahead_quiesce(c->device); g_tp_kv_epoch++; }
    // While a step runs ahead the device already holds the scalars that step needs, and ggml-alloc may have handed their memory to later nodes of the
    // same graph (the token id's block becomes part of the logits): a write now would land in the middle of -- or after -- the run and clobber it.
    // The value is recorded above; graph_compute either finds it equal to the prediction (nothing to write) or writes all of them before the real run.
    else if (c->device < 64 && g_ahead_ctx[c->device] && g_ahead_ctx[c->device]->ahead.inflight) return;
    if (size <= g_ring.max && ring_set(g_ring, c->device, (char *) t->data + off, data, size)) return;
    if (!upload(c->device, (char *) t->data + off, data, size)) GGML_LOG_ERROR("[ggml-hip] set_tensor '%s' (%zu bytes at %zu) failed: %s\n", t->name, size, off, cllm_last_error());
}
// get_tensor (the logits of every token: 513 KB for Llama-3): through a page-locked staging area -- the D2H copy into the host's pageable
// destination is several times slower than DMA into pinned memory + a memcpy
int tp_ahead_replay(hip_backend_ctx * c);
void * g_stage = nullptr; size_t g_stage_size = 0;
constexpr size_t k_stage_chunk = 4u << 20;
std::mutex g_stage_mutex;
void ahead_launch(hip_backend_ctx * c, bool chained = false);
void buf_get_impl(ggml_backend_buffer_t b, const char * src, const char * name, void * data, size_t size);
void buf_get(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t off, size_t size) {
    ws_scope ws(g_ws.get_us); g_ws.gets++;
    flush_sets();
    const int device = ((hip_buffer_ctx *) b->context)->device;
    hip_backend_ctx * ac = device >= 0 && device < 64 ? g_ahead_ctx[device] : nullptr;
    const char * src = (const char *) t->data + off;
    // the host reads the logits: the next step is started FIRST (it snapshots the graph's outputs on its stream, then runs), and the 513 KB copy to the host below is
    // served from that snapshot while the step already computes -- started after the copy, the GPU idled for the copy's ~70 us of every token
    static const bool ahead_late = getenv("CLLM_HIP_AHEAD_LATE") != nullptr;       // (A/B switch: round 3's order)
    const bool is_logits_read = ac && ac->ahead.armed && !ac->ahead.inflight && (const void *) t->data == ac->ahead.logits_ptr && off == 0 && size == ac->ahead.logits_bytes;
    if (is_logits_read && !ahead_late) ahead_launch(ac);
    if (ac && (ac->ahead.inflight || (is_logits_read && ac->ahead.snap_ready))) {
        bool served = false;
        if (ac->ahead.inflight) ahead_finish_event(ac);      // (a host that reads without synchronize(): the snapshot must be complete)
        for (const auto & o : ac->ahead.outs)      // the step running ahead overwrites the graph's outputs: their snapshots
            if (src >= o.ptr && src + size <= o.ptr + o.bytes) { src = (const char *) ac->ahead.snap + o.snap_off + (src - o.ptr); served = true; break; }
        if (!served && ac->ahead.inflight) ahead_quiesce(device);        // anything else: what the step running ahead leaves behind
        ac->ahead.snap_ready = false;
    }
    buf_get_impl(b, src, t->name, data, size);
    if (is_logits_read && ahead_late && ac->ahead.armed && !ac->ahead.inflight) ahead_launch(ac);
}
void buf_get_impl(ggml_backend_buffer_t b, const char * src, const char * name, void * data, size_t size) {
    std::lock_guard<std::mutex> lock(g_stage_mutex);            // one staging area (accessory models may read from other threads)
    cllm_set_device(((hip_buffer_ctx *) b->context)->device);
    const size_t want = size < k_stage_chunk ? size : k_stage_chunk;
    if (size >= 4096 && g_stage_size < want) {
        if (g_stage) cllm_host_free(g_stage);
        g_stage = nullptr; g_stage_size = 0;
        if (cllm_host_malloc(&g_stage, k_stage_chunk) == CLLM_OK) g_stage_size = k_stage_chunk;
    }
    if (size < 4096 || !g_stage) {
        if (cllm_memcpy_d2h(data, src, size, nullptr) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] get_tensor '%s' failed: %s\n", name, cllm_last_error());
        return;
    }
    for (size_t done = 0; done < size; done += g_stage_size) {
        const size_t n = size - done < g_stage_size ? size - done : g_stage_size;
        if (cllm_memcpy_d2h(g_stage, src + done, n, nullptr) != CLLM_OK) {     // synchronous
            GGML_LOG_ERROR("[ggml-hip] get_tensor '%s' failed: %s\n", name, cllm_last_error());
            return;
        }
        memcpy((char *) data + done, g_stage, n);
    }
}
bool buf_cpy(ggml_backend_buffer_t b, const ggml_tensor * src, ggml_tensor * dst) {
    if (!ggml_is_contiguous(src) || !ggml_is_contiguous(dst) || ggml_nbytes(src) != ggml_nbytes(dst)) return false;
    flush_sets();
    ahead_quiesce(((hip_buffer_ctx *) b->context)->device);
    if (src->buffer && src->buffer->iface.get_base == buf_base) ahead_quiesce(((hip_buffer_ctx *) src->buffer->context)->device);
    cllm_set_device(((hip_buffer_ctx *) b->context)->device); ((hip_buffer_ctx *) b->context)->gen++; g_tp_kv_epoch++;
    { std::lock_guard<std::mutex> lock(g_ring.m); i32_forget(dst->data, ggml_nbytes(dst)); }
    if (ggml_backend_buffer_is_host(src->buffer)) {
        if (cllm_memcpy_h2d(dst->data, src->data, ggml_nbytes(src), nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) { GGML_LOG_ERROR("[ggml-hip] cpy_tensor (host -> '%s') failed: %s\n", dst->name, cllm_last_error()); return false; }
        return true;
    }
    if (src->buffer && src->buffer->iface.get_base == buf_base) {     // another buffer of this module (any device: peer access through hipMemcpy)
        if (cllm_memcpy_d2d(dst->data, src->data, ggml_nbytes(src), nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) { GGML_LOG_ERROR("[ggml-hip] cpy_tensor ('%s' -> '%s') failed: %s\n", src->name, dst->name, cllm_last_error()); return false; }
        return true;
    }
    return false;
}
void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    auto * c = (hip_buffer_ctx *) b->context; ahead_quiesce(c->device); cllm_set_device(c->device); c->gen++; g_tp_kv_epoch++;
    { std::lock_guard<std::mutex> lock(g_ring.m); i32_forget(c->base, b->size); }
    if (cllm_memset(c->base, v, b->size, nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] clear failed: %s\n", cllm_last_error());
}
const ggml_backend_buffer_i k_buffer_i = { buf_free, buf_base, nullptr, buf_memset, buf_set, buf_get, buf_cpy, buf_clear, nullptr };

// ---------------------------------------------------------------------------------------------------------------------------
// packed weights: mat-vecs that read the same activation become ONE launch over a row-repacked copy of their weight matrices
//   q | k | v concatenated (the three projections of an attention block), gate / up with alternating rows (SiLU(gate)*up in the epilogue).
// The copy is made on the device the first time the pattern is seen (cllm_pack_rows) and stays valid as long as none of the source
// buffers is written through the buffer interface (generation counters) or freed.  It costs HBM -- the originals belong to the host and
// stay -- so the total is capped: CLLM_HIP_PACK_GB (default: a quarter of the device), and a tenth of the device must stay free.
// CLLM_HIP_PACK=0 turns it off.
// ---------------------------------------------------------------------------------------------------------------------------
struct pack_entry { void * data = nullptr; size_t bytes = 0; int n = 0; int device = 0; const void * src[3] = {}; uint64_t uid[3] = {}, gen[3] = {}; bool refused = false; };
std::unordered_map<const void *, pack_entry> g_packs;          // keyed by the first source's data pointer
size_t g_pack_bytes = 0;
std::mutex g_pack_mutex;                                        // (accessory models own their own contexts and may run on other threads)
void packs_forget(uint64_t uid) {
    std::lock_guard<std::mutex> lock(g_pack_mutex);
    for (auto it = g_packs.begin(); it != g_packs.end();) {
        bool hit = false;
        for (int i = 0; i < it->second.n; i++) hit = hit || it->second.uid[i] == uid;
        if (hit) { if (it->second.data) { cllm_free(it->second.data); g_pack_bytes -= it->second.bytes; } it = g_packs.erase(it); } else ++it;
    }
}
bool ours(const ggml_tensor * t) { return t && t->buffer && t->buffer->iface.free_buffer == buf_free && !t->view_src; }
// the packed copy of n (2 or 3) same-type, same-row-length weight matrices, or nullptr (not ours / over budget / allocation failed)
void * get_pack(int device, void * stream, const ggml_tensor * const * w, int n, bool interleave, bool flat = false) {      // flat: byte blobs (bias vectors)
    static const bool off = getenv("CLLM_HIP_PACK") && atoi(getenv("CLLM_HIP_PACK")) == 0;
    if (off) return nullptr;
    for (int i = 0; i < n; i++) if (!ours(w[i]) || ((const hip_buffer_ctx *) w[i]->buffer->context)->device != device) return nullptr;
    std::lock_guard<std::mutex> lock(g_pack_mutex);
    pack_entry & e = g_packs[w[0]->data];
    bool valid = e.n == n && e.device == device;
    for (int i = 0; i < n && valid; i++) {
        const auto * bc = (const hip_buffer_ctx *) w[i]->buffer->context;
        valid = e.src[i] == w[i]->data && e.uid[i] == bc->uid && e.gen[i] == bc->gen.load();
    }
    if (valid) return e.refused ? nullptr : e.data;
    if (e.data) { cllm_stream_sync(stream); cllm_free(e.data); g_pack_bytes -= e.bytes; }
    e = pack_entry(); e.n = n; e.device = device; e.refused = true;
    size_t bytes = 0; int64_t rows[3]; const void * srcs[3];
    for (int i = 0; i < n; i++) {
        const auto * bc = (const hip_buffer_ctx *) w[i]->buffer->context;
        e.src[i] = w[i]->data; e.uid[i] = bc->uid; e.gen[i] = bc->gen.load();
        if (!flat && w[i]->ne[2] > 1 && w[i]->nb[2] != (size_t) w[i]->ne[1] * w[i]->nb[1]) return nullptr;      // expert slabs must be contiguous
        rows[i] = flat ? (int64_t) ggml_nbytes(w[i]) : w[i]->ne[1] * w[i]->ne[2]; srcs[i] = w[i]->data; bytes += flat ? ggml_nbytes(w[i]) : (size_t) rows[i] * w[i]->nb[1];
    }
    size_t mfree = 0, mtotal = 0;
    cllm_device_info(device, nullptr, 0, &mfree, &mtotal, nullptr);
    static const double cap_gb = getenv("CLLM_HIP_PACK_GB") ? atof(getenv("CLLM_HIP_PACK_GB")) : -1.0;
    const size_t cap = cap_gb >= 0 ? (size_t)(cap_gb * 1e9) : mtotal / 4;
    if (g_pack_bytes + bytes > cap || mfree < bytes + mtotal / 10) return nullptr;
    void * p = nullptr;
    if (cllm_malloc(&p, bytes) != CLLM_OK) return nullptr;
    if (cllm_pack_rows(stream, p, srcs, rows, n, flat ? 1 : w[0]->nb[1], interleave ? 1 : 0) != CLLM_OK || cllm_stream_sync(stream) != CLLM_OK) { cllm_free(p); return nullptr; }
    e.data = p; e.bytes = bytes; e.refused = false; g_pack_bytes += bytes;
    return p;
}

// ---------------------------------------------------------------------------------------------------------------------------
// buffer type
// ---------------------------------------------------------------------------------------------------------------------------
const char * buft_name(ggml_backend_buffer_type_t t) { return ((hip_device_ctx *) t->context)->name.c_str(); }
ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    ws_scope ws(g_ws.alloc_us); g_ws.allocs++;
    auto * d = (hip_device_ctx *) t->context;
    cllm_set_device(d->id);
    void * p = nullptr;
    if (cllm_malloc(&p, size ? size : 1) != CLLM_OK) { HIPB_LOG("alloc of %zu bytes failed: %s", size, cllm_last_error()); return nullptr; }
    return ggml_backend_buffer_init(t, k_buffer_i, new hip_buffer_ctx{ d->id, p, g_next_buffer_uid.fetch_add(1) }, size);
}
size_t buft_align(ggml_backend_buffer_type_t) { return 256; }
bool buft_is_host(ggml_backend_buffer_type_t) { return false; }
const ggml_backend_buffer_type_i k_buft_i = { buft_name, buft_alloc, buft_align, nullptr, nullptr, buft_is_host };

// ---------------------------------------------------------------------------------------------------------------------------
// supports_op: exactly the combinations cllm_op_* accept (anything else goes to the CPU backend through the scheduler)
// ---------------------------------------------------------------------------------------------------------------------------
bool dev_supports_op(ggml_backend_dev_t, const ggml_tensor * op) {
    const ggml_tensor * a = op->src[0], * b = op->src[1];
    switch (op->op) {
        case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: return true;
        case GGML_OP_MUL_MAT:
            if (!f32_dense(b) || op->type != GGML_TYPE_F32 || !dense_rows(a)) return false;
            // (the predicates of cllm_op_mul_mat / check_mm that do not depend on the data pointers: what is declined here runs on the CPU backend
            // instead of failing in graph_compute; buffers of this module are 256-byte aligned and ggml-alloc keeps that alignment)
            if (is_q(a->type)) return a->ne[0] % 32 == 0 && b->nb[1] % 16 == 0 && b->nb[2] % 16 == 0 && b->nb[3] % 16 == 0 &&
                                      (a->type != GGML_TYPE_Q4_K || (a->nb[1] % 16 == 0 && a->nb[2] % 16 == 0 && a->nb[3] % 16 == 0)) &&
                                      (a->type != GGML_TYPE_Q4_1 || (a->nb[1] % 4 == 0 && a->nb[2] % 4 == 0 && a->nb[3] % 4 == 0));
            if (is_kq(a->type)) return a->ne[0] % ggml_blck_size(a->type) == 0 && b->nb[1] % 16 == 0 && b->nb[2] % 16 == 0 && b->nb[3] % 16 == 0 && (a->type != GGML_TYPE_Q5_K || a->nb[1] % 16 == 0) &&
                                       (a->type == GGML_TYPE_MXFP4 || a->nb[1] % 2 == 0);
            return (a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32) && a->ne[2] * a->ne[3] <= 65535;
        case GGML_OP_MUL_MAT_ID:
            if (is_kq(a->type))        // the coverage types: one grid slice per (token, slot) of gemv_kq.hip (no fused MoE launch takes them)
                return f32_dense(b) && op->src[2] && op->src[2]->type == GGML_TYPE_I32 && a->ne[3] == 1 && b->ne[3] == 1 && a->ne[0] % ggml_blck_size(a->type) == 0 && dense_rows(a) &&
                       (b->ne[1] == 1 || b->ne[1] == op->src[2]->ne[0]) && op->src[2]->ne[0] * op->src[2]->ne[1] <= 65535 && b->nb[1] % 16 == 0 && b->nb[2] % 16 == 0 &&
                       (a->type != GGML_TYPE_Q5_K || (a->nb[1] % 16 == 0 && a->nb[2] % 16 == 0)) && (a->type == GGML_TYPE_MXFP4 || (a->nb[1] % 2 == 0 && a->nb[2] % 2 == 0));
            return is_q(a->type) && f32_dense(b) && op->src[2] && op->src[2]->type == GGML_TYPE_I32 && a->ne[3] == 1 && b->ne[3] == 1 && a->ne[0] % 32 == 0 &&
                   (b->ne[1] == 1 || b->ne[1] == op->src[2]->ne[0]) && op->src[2]->ne[0] * op->src[2]->ne[1] <= 65535 && b->nb[1] % 16 == 0 && b->nb[2] % 16 == 0 &&
                   (a->type != GGML_TYPE_Q4_K || (a->nb[1] % 16 == 0 && a->nb[2] % 16 == 0)) && (a->type != GGML_TYPE_Q4_1 || (a->nb[1] % 4 == 0 && a->nb[2] % 4 == 0));
        case GGML_OP_RMS_NORM: return f32_dense(a) && op->type == GGML_TYPE_F32;
        case GGML_OP_ADD: case GGML_OP_MUL: case GGML_OP_DIV: return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && ggml_can_repeat(b, a);
        case GGML_OP_SUM_ROWS: return f32_dense(a) && f32_dense(op);
        case GGML_OP_TOP_K: return f32_dense(a) && op->type == GGML_TYPE_I32 && a->ne[0] <= 4096;
        case GGML_OP_SCALE: case GGML_OP_DIAG_MASK_INF: return a->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32;
        case GGML_OP_UNARY: return ggml_get_unary_op(op) == GGML_UNARY_OP_SILU && f32_dense(a) && f32_dense(op);
        case GGML_OP_ROPE: {
            const int mode = op->op_params[2];
            return f32_dense(a) && f32_dense(op) && b && b->type == GGML_TYPE_I32 && (mode == 0 || mode == 2) && (!op->src[2] || op->src[2]->type == GGML_TYPE_F32);
        }
        case GGML_OP_SOFT_MAX: {
            float max_bias; memcpy(&max_bias, (const float *) op->op_params + 1, 4);
            return f32_dense(a) && f32_dense(op) && max_bias == 0.0f && !op->src[2] && (!b || b->type == GGML_TYPE_F32 || b->type == GGML_TYPE_F16);
        }
        case GGML_OP_SET_ROWS:
            if (op->type == GGML_TYPE_Q8_0) return f32_dense(a) && a->ne[0] % 32 == 0 && a->nb[1] % 16 == 0 && a->nb[2] % 16 == 0 && a->nb[3] % 16 == 0 && (b->type == GGML_TYPE_I32 || b->type == GGML_TYPE_I64);
            return f32_dense(a) && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) && (b->type == GGML_TYPE_I32 || b->type == GGML_TYPE_I64);
        case GGML_OP_FLASH_ATTN_EXT: {      // what cllm_op_flash_attn_ext takes (`-fa`, src/layers.cpp:2634-2656); anything else stays on the CPU backend
            const ggml_tensor * v = op->src[2], * m = op->src[3];
            float max_bias, softcap; memcpy(&max_bias, (const float *) op->op_params + 1, 4); memcpy(&softcap, (const float *) op->op_params + 2, 4);
            if (!a || !b || !v || op->src[4] || max_bias != 0.0f || softcap != 0.0f || op->type != GGML_TYPE_F32 || a->type != GGML_TYPE_F32 || a->nb[0] != 4) return false;
            if (b->type != v->type || (b->type != GGML_TYPE_F16 && b->type != GGML_TYPE_Q8_0) || b->nb[0] != ggml_type_size(b->type) || v->nb[0] != ggml_type_size(v->type)) return false;
            const int64_t D = a->ne[0];
            if ((D != 64 && D != 128) || b->ne[0] != D || v->ne[0] != D || v->ne[1] != b->ne[1] || b->ne[2] <= 0 || a->ne[2] % b->ne[2] || v->ne[2] != b->ne[2] || a->ne[3] != b->ne[3] || a->ne[3] != v->ne[3]) return false;
            if (a->nb[1] % 16 || a->nb[2] % 16 || a->nb[3] % 16 || !ggml_is_contiguous(op)) return false;
            if (b->type == GGML_TYPE_F16 && (b->nb[1] % 16 || b->nb[2] % 16 || b->nb[3] % 16 || v->nb[1] % 16 || v->nb[2] % 16 || v->nb[3] % 16)) return false;
            if (m && (m->type != GGML_TYPE_F16 || !ggml_is_contiguous(m) || m->ne[0] < b->ne[1] || m->ne[1] < a->ne[1])) return false;
            // the C ABI's own predicates (cllm_op_flash_attn_ext): a mask broadcasts over heads / batches only as 1 or the full count (ggml allows any divisor)
            if (m && ((m->ne[2] != 1 && m->ne[2] != a->ne[2]) || (m->ne[3] != 1 && m->ne[3] != a->ne[3]))) return false;
            // (q's data pointer must be 16-byte aligned as well: every block ggml-alloc hands out is aligned to the buffer type's 256 bytes and the reference's
            //  q is a PERMUTE of a contiguous projection -- checked again at run time, where a mismatch is an error rather than a fallback)
            return a->ne[2] <= 65535 && a->ne[3] <= 65535;
        }
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const ggml_type s = a->type, d = op->type;
            const bool fs = s == GGML_TYPE_F32 || s == GGML_TYPE_F16, fd = d == GGML_TYPE_F32 || d == GGML_TYPE_F16;
            if (s == d && is_q(s)) return a->ne[0] == op->ne[0] && a->nb[0] == ggml_type_size(s) && op->nb[0] == ggml_type_size(s);      // whole quantized rows (CONT of the Q8_0 K-cache view)
            return (fs && fd) || (s == GGML_TYPE_I32 && d == GGML_TYPE_I32);
        }
        case GGML_OP_GET_ROWS: return (is_q(a->type) || is_kq(a->type) || a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32) && b->type == GGML_TYPE_I32 && op->type == GGML_TYPE_F32;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backend (stream): graph_compute
// ---------------------------------------------------------------------------------------------------------------------------
void tp_free_all();
bool tp_check_errors();
struct fuse_plan;
void ahead_finish_event(hip_backend_ctx * c);
const char * be_name(ggml_backend_t b) { return ((hip_device_ctx *) b->device->context)->name.c_str(); }      // (the ggml device's name: several of them can sit on one GPU, CLLM_HIP_VIRTUAL_DEVICES)
void be_free(ggml_backend_t b) {
    auto * c = (hip_backend_ctx *) b->context;
    cllm_set_device(c->device); cllm_stream_sync(c->stream);
    if (c->wdata) cllm_free(c->wdata);
    if (c->abuf) cllm_free(c->abuf);
    if (c->graph_exec) cllm_graph_destroy(c->graph_exec);
    if (c->copy_event) cllm_event_destroy(c->copy_event);
    if (c->device < 64 && g_ahead_ctx[c->device] == c) g_ahead_ctx[c->device] = nullptr;
    if (g_tp.owner == c) { tp_free_all(); cllm_set_device(c->device); }      // (tensor parallel: the ranks' streams, shards and receive buffers hang off this backend's stream)
    if (c->ahead.ev) cllm_event_destroy(c->ahead.ev);
    if (c->ahead.tok_host) cllm_host_free(c->ahead.tok_host);
    if (c->ahead.table_dev) cllm_free(c->ahead.table_dev);
    if (c->ahead.tab_host) cllm_host_free(c->ahead.tab_host);
    if (c->ahead.scratch) cllm_free(c->ahead.scratch);
    if (c->ahead.rng_dev) cllm_free(c->ahead.rng_dev);
    if (c->ahead.rng_host) cllm_host_free(c->ahead.rng_host);
    if (c->ahead.snap) cllm_free(c->ahead.snap);
    cllm_stream_destroy(c->stream);
    delete c; delete b;
}
void be_sync(ggml_backend_t b) {
    ws_scope ws(g_ws.sync_us); g_ws.syncs++;
    flush_sets();
    auto * c = (hip_backend_ctx *) b->context; cllm_set_device(c->device);
    // a step queued behind the one the host waits for (ahead_launch chained): the host's step is complete -- outputs snapshotted -- when the event behind its arg-max has fired
    if (c->ahead.inflight && c->ahead.chained) ahead_finish_event(c); else cllm_stream_sync(c->stream);
    cllm_stream_sync(nullptr);
    // a bounded wait inside a kernel that timed out leaves void results behind: synchronize() cannot return a status (ggml-backend-impl.h:96), so it is said
    // loudly here and the next graph_compute of this backend fails
    if (cllm_check_kernel_errors() != CLLM_OK) { GGML_LOG_ERROR("[ggml-hip] %s\n", cllm_last_error()); c->kernel_error = true; }
    if (!tp_check_errors()) { GGML_LOG_ERROR("[ggml-hip] tensor parallel: a granule wait of the fused all-reduce timed out (a rank is late, dead or out of step): the step's logits are void\n"); c->kernel_error = true; }
}

int ensure_wdata(hip_backend_ctx * c, size_t need) {
    if (need <= c->wsize) return CLLM_OK;
    cllm_stream_sync(c->stream);                    // kernels in flight may still read the old scratch
    if (c->wdata) cllm_free(c->wdata);
    c->wdata = nullptr; c->wsize = 0;
    const size_t sz = need + need / 4 + 4096;
    if (int rc = cllm_malloc(&c->wdata, sz)) return rc;
    c->wsize = sz;
    return CLLM_OK;
}

int ensure_abuf(hip_backend_ctx * c, size_t need) {      // fused attention: cos/sin table | q,k,v projections | long-context scores
    if (need <= c->asize) return CLLM_OK;
    cllm_stream_sync(c->stream);
    if (c->abuf) cllm_free(c->abuf);
    c->abuf = nullptr; c->asize = 0;
    const size_t sz = need + need / 4 + 4096;
    if (int rc = cllm_malloc(&c->abuf, sz)) return rc;
    c->asize = sz;
    return CLLM_OK;
}

// ---- node-pattern fusion (decode graphs: a token is ~25 launches per layer node by node, ~5 us each) -------------------------------
// A pattern is fused only if every intermediate it swallows is used by nobody else: the graph-wide use count
// (ggml_node_get_use_count, shared by scheduler splits) must equal the uses found in THIS graph, and the tensor must not be a
// graph output.  All fused launches are bit-identical to the node sequence they replace (same kernels' arithmetic).
//   RMS_NORM -> MUL(weight) -> {MUL_MAT ...}      norm + weight + quantize in every consumer mat-vec's prologue (pro 1)
//   UNARY(SILU) -> MUL(up) -> MUL_MAT             SiLU*up + quantize in the mat-vec's prologue (pro 4)
//   MUL_MAT -> ADD(residual | bias)               epilogue
//   SCALE -> DIAG_MASK_INF -> SOFT_MAX            cllm_op_scale_mask_soft_max
//   {MUL_MAT q, k, v} of a fused attention block      one launch over the packed q|k|v weights (get_pack)
//   {MUL_MAT gate, up} -> SILU -> MUL -> MUL_MAT       one launch over the row-interleaved gate/up weights with the SiLU*up epilogue; the
//                                                      consumer mat-vec then only quantizes (pro 2)
struct fused_mv {
    int pro = 2; const float * px = nullptr; const float * pw = nullptr; float eps = 0.0f; const float * resid = nullptr; float * dst = nullptr;
    const ggml_tensor * resid_t = nullptr;      // the ADD's other operand (a bias when it is a weight leaf)
    int node = -1;                      // the MUL_MAT node
    int ga = -1, gb = -1;               // pro 4: the nodes producing px (gate) and pw (up)
    int group = -1;                     // member of a merged launch (fuse_plan::groups)
    bool alias = false;                 // dst overlaps an input every workgroup reads in its prologue (ggml-alloc re-used a freed parent's block): stage the output
};
struct merge_group {
    int n = 0, member[3] = { -1, -1, -1 };      // entries of mvs, in packed row order
    bool interleave = false;                     // gate/up: SiLU*up epilogue, output = half the rows
    bool bias = false;                           // q|k|v with biases (Qwen2): the three bias vectors are packed too and ride in the epilogue
    int consumer = -1;                           // gate/up: the entry of mvs (pro 4) that reads the activation
    int state = 0;                               // at run time: 0 not reached, 1 launched merged, 2 members launch separately
};
//   ROPE(q) ROPE(k) SET_ROWS(k) CPY(v) MUL_MAT(K,Q) SCALE DIAG_MASK_INF SOFT_MAX MUL_MAT(V,P) PERMUTE CONT
//                                                 the single-token attention block of KVCacheAttention (src/layers.cpp:3044-3123,
//                                                 2499-2561) as cllm_op_rope_kv_attn_decode (level 2: the q/k/v mat-vecs write into
//                                                 module scratch) or, with a RoPE flavour that call does not take (YaRN, frequency
//                                                 factors, partial rotation), the last seven as cllm_op_attn_decode (level 1)
struct fused_attn {
    int level = 0;
    int wq = -1, wk = -1, wv = -1;      // level 2: entries of mvs producing the un-rotated q / k / v
    const float * q = nullptr;          // level 1: the rotated q
    const int32_t * pos = nullptr;      // I32 [1] on the device: the position == cached length - 1
    bool alias = false;                 // level 1: out overlaps the rotated q other heads still read: stage the output
    int nh = 0, nkv = 0, hd = 0, mode = 0;
    float freq_base = 0.0f;
    int64_t n_kv = 0, ML = 0;
    void * k_cache = nullptr, * v_cache = nullptr;
    float * out = nullptr;
};
//   UNARY(SILU) -> MUL                               cllm_op_silu_mul (consumers other than a single-column mat-vec: the expert block of a sparse MoE)
//   RMS_NORM -> MUL(weight)                          cllm_op_rms_norm_mul (where the norm cannot go into its consumers' prologues)
//   GET_ROWS(probs, ids) -> SUM_ROWS -> DIV -> MUL(experts) -> ADD of the slot views (-> ADD residual)
//                                                    cllm_op_moe_combine: the tail of GenericSparseMLP::forward (src/layers.cpp:3792-3872)
//   {MUL_MAT_ID gate, MUL_MAT_ID up} -> UNARY(SILU) -> MUL     one token: cllm_op_mul_mat_id_silu_mul over the per-expert interleaved pack
struct fused_moe { const ggml_tensor * experts = nullptr, * probs = nullptr, * ids = nullptr, * resid = nullptr; int down = -1; /* the MUL_MAT_ID node folded into the launch */ };
//   RMS_NORM -> MUL(weight) -> {MUL_MAT(router) -> SOFT_MAX -> TOP_K, experts}     one token: cllm_op_moe_router (launched at the TOP_K node)
struct moe_router { const ggml_tensor * x = nullptr, * w = nullptr, * gate = nullptr, * xnorm = nullptr, * probs = nullptr; float eps = 0;
                    int itk = -1; std::vector<int> xn_users, out_users; };      // itk: the TOP_K node; who else reads the normalised activation / the probabilities and ids (end nodes behind no-op views)
//   prefill (>= cllm_mul_mat_ex_min_cols() columns): RMS_NORM -> MUL -> {MUL_MAT ...} (norm in the quantizer; further projections of the same activation reuse
//   the act rows), UNARY(SILU) -> MUL -> MUL_MAT (SiLU * up in the quantizer), MUL_MAT -> ADD (residual in the epilogue): cllm_op_mul_mat_ex at the MUL_MAT node
struct pf_mm { int pro = 0; const ggml_tensor * x = nullptr, * w2 = nullptr; float eps = 0; const ggml_tensor * resid = nullptr, * out = nullptr; };
struct moe_gate_up { int mul = -1, gate = -1, up = -1, unary = -1; void * W = nullptr; int router = -1; };      // node indices; W: the pack, resolved before the walk;
                                                                                                                  // router: the block's router folded into this launch (cllm_op_moe_router_gate_up)
//   MUL_MAT(K, Q) SCALE DIAG_MASK_INF SOFT_MAX MUL_MAT(V^T, P)   with more than 32 query rows (the tolerance tier of the MFMA mat-muls):
//                                                    cllm_op_attn_prefill, one flash kernel in place of the V.P node; the scores never reach HBM
struct fused_fa { int ikq = -1, n_past = 0; float scale = 1.0f; bool alias = false; size_t bytes = 0; };   // alias: dst overlaps q (ggml-alloc reuses the dead q block): staged
enum { ALT_NONE = 0, ALT_SILU_MUL = 1 /* src0 is the SiLU */, ALT_RMS_NORM_MUL = 2, ALT_MOE_COMBINE = 3, ALT_MUL_SILU = 4 /* src1 is the SiLU */, ALT_MOE_GATE_UP = 5, ALT_FLASH_PREFILL = 6, ALT_MOE_ROUTER = 7 };
struct fuse_plan {
    std::vector<uint8_t> skip;          // node is produced inside a fused launch (or not needed at all)
    std::vector<uint8_t> alt;           // the node is launched as one of the ALT_* forms
    std::vector<int>     moe;           // ALT_MOE_COMBINE: index into moes; ALT_FLASH_PREFILL: index into fas
    std::vector<fused_fa> fas;
    std::vector<fused_moe> moes;
    std::vector<moe_router> routers;    // ALT_MOE_ROUTER: P.moe[top_k node] indexes them
    std::vector<pf_mm> pfs; std::vector<int> pf;       // prefill mat-muls with a fused prologue / epilogue: pf[MUL_MAT node] indexes pfs
    std::vector<moe_gate_up> gus;       // candidates (P.moe[mul node] indexes them once resolved)
    std::vector<int>     mv;            // index into mvs for MUL_MAT nodes launched fused, else -1
    std::vector<fused_mv> mvs;
    std::vector<int>     sm_src;        // SOFT_MAX nodes: node index of the SCALE feeding the fused scale+mask+soft_max, else -1
    std::vector<int>     attn;          // CONT nodes: index into attns of the fused attention launched in their place, else -1
    std::vector<fused_attn> attns;
    std::vector<merge_group> groups;
};

bool f32_vec(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->nb[0] == 4 && t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && ((uintptr_t) t->data & 15) == 0; }
bool overlap(const void * a, size_t na, const void * b, size_t nb) { return (const char *) a < (const char *) b + nb && (const char *) b < (const char *) a + na; }

// planning runs once per graph_compute, before the first launch: flat, allocation-free containers (reused between calls)
struct int_span {
    const int * b, * e;
    const int * begin() const { return b; }
    const int * end() const { return e; }
    int operator[](size_t k) const { return b[k]; }
};
struct user_lists {                 // the consumers of every node, in node order (CSR)
    std::vector<int> start, list;
    int_span operator[](int i) const { return { list.data() + start[i], list.data() + start[i + 1] }; }
};
struct node_index {                 // ggml_tensor * -> position in the graph (open addressing)
    std::vector<const ggml_tensor *> key; std::vector<int> val; size_t mask = 0;
    static size_t h(const void * p) { return (size_t)(((uintptr_t) p >> 4) * 0x9E3779B97F4A7C15ull >> 24); }
    void build(ggml_cgraph * g, int n) {
        size_t cap = 64; while (cap < (size_t) n * 2) cap <<= 1;
        key.assign(cap, nullptr); val.resize(cap); mask = cap - 1;
        for (int i = 0; i < n; i++) { const ggml_tensor * t = ggml_graph_node(g, i); size_t s = h(t) & mask; while (key[s]) s = (s + 1) & mask; key[s] = t; val[s] = i; }
    }
    int operator()(const ggml_tensor * t) const { for (size_t s = h(t) & mask;; s = (s + 1) & mask) { if (key[s] == t) return val[s]; if (!key[s]) return -1; } }
};

void plan_attention(ggml_cgraph * g, fuse_plan & P, const std::vector<int> & local, const std::vector<int> & writer,
                    const user_lists & users, const node_index & find) {
    static const int max_level = getenv("CLLM_HIP_FUSE_ATTN") ? atoi(getenv("CLLM_HIP_FUSE_ATTN")) : 2;
    if (max_level <= 0) return;
    const int n = ggml_graph_n_nodes(g);
    auto node = [&](int i) { return ggml_graph_node(g, i); };
    auto only_local = [&](int i, int uses) {
        return i >= 0 && !(node(i)->flags & GGML_TENSOR_FLAG_OUTPUT) && local[i] == uses && ggml_node_get_use_count(g, i) == uses;
    };
    static const int flash_min = cllm_attn_prefill_min_cols();
    // x -> RESHAPE* -> ROPE -> RESHAPE* -> end : returns the ROPE and the node under it; every node of the chain is collected
    auto through_reshapes = [&](const ggml_tensor * t, std::vector<int> & chain) {      // (a VIEW of everything at offset 0 is a reshape too)
        while (t && (t->op == GGML_OP_RESHAPE || (t->op == GGML_OP_VIEW && t->src[0] && t->data == t->src[0]->data && ggml_nelements(t) == ggml_nelements(t->src[0]) &&
                                                   ggml_is_contiguous(t) && ggml_is_contiguous(t->src[0])))) { chain.push_back(find(t)); t = t->src[0]; }
        return t;
    };
    for (int i = 0; i < n; i++) {
        if (P.sm_src[i] < 0) continue;
        const ggml_tensor * sm = node(i), * scn = node(P.sm_src[i]);
        const ggml_tensor * kq = scn->src[0];
        const int ikq = find(kq);
        if (ikq < 0 || kq->op != GGML_OP_MUL_MAT || !only_local(ikq, 1) || !only_local(i, 1)) continue;
        const ggml_tensor * kp = kq->src[0], * qp = kq->src[1];
        if (kp->type == GGML_TYPE_F16 && qp->type == GGML_TYPE_F32 && qp->ne[1] >= flash_min && users[i].end() - users[i].begin() == 1) {      // ---- prefill: the flash kernel
            const int ikqv = users[i][0];
            const ggml_tensor * kqv = node(ikqv), * vv = kqv->src[0];
            const int64_t hd = kp->ne[0], n_kv = kp->ne[1], nkv = kp->ne[2], nh = qp->ne[2], ql = qp->ne[1];
            const int n_past = sm->src[0]->op_params[0];
            auto al16 = [](const ggml_tensor * t) { return (((uintptr_t) t->data | t->nb[1] | t->nb[2] | t->nb[3]) & 15) == 0; };
            if (kqv->op == GGML_OP_MUL_MAT && kqv->src[1] == sm && P.mv[ikqv] < 0 && !P.skip[ikqv] && (hd == 64 || hd == 128) && kp->ne[3] == 1 && qp->ne[3] == 1 && qp->ne[0] == hd &&
                nkv > 0 && nh % nkv == 0 && n_past >= 0 && n_kv == n_past + ql && kp->nb[0] == 2 && qp->nb[0] == 4 && al16(kp) && al16(qp) && al16(kqv) && kqv->type == GGML_TYPE_F32 && kqv->nb[0] == 4 &&
                vv->type == GGML_TYPE_F16 && vv->ne[0] == n_kv && vv->ne[1] == hd && vv->ne[2] == nkv && vv->ne[3] == 1 && vv->nb[0] == 2 && al16(vv) && (int64_t)(vv->nb[1] / 2) >= n_kv &&
                ggml_is_contiguous(kqv) && sm->src[0]->src[0] == scn) {
                fused_fa F; F.ikq = ikq; F.n_past = n_past; memcpy(&F.scale, scn->op_params, 4);
                static const bool force_stage = getenv("CLLM_HIP_FORCE_STAGE") != nullptr;
                size_t qext = 0;                            // the bytes the (permuted) q view spans
                for (int k = 0; k < 4; k++) qext += (size_t)(qp->ne[k] - 1) * qp->nb[k];
                F.bytes = ggml_nbytes(kqv);
                F.alias = force_stage || overlap(kqv->data, F.bytes, qp->data, qext + 4);
                P.skip[ikq] = P.skip[i] = 1;
                P.alt[ikqv] = ALT_FLASH_PREFILL; P.moe[ikqv] = (int) P.fas.size(); P.fas.push_back(F);
            }
            continue;
        }
        if (kp->type != GGML_TYPE_F16 || qp->type != GGML_TYPE_F32 || qp->op != GGML_OP_PERMUTE) continue;
        const int64_t hd = kp->ne[0], n_kv = kp->ne[1], nkv = kp->ne[2], nh = qp->ne[2];
        if (kp->ne[3] != 1 || qp->ne[0] != hd || qp->ne[1] != 1 || qp->ne[3] != 1 || nkv <= 0 || nh <= 0 || nh % nkv || n_kv < 1) continue;
        const size_t KD = (size_t) hd * nkv;
        if (kp->nb[0] != 2 || kp->nb[1] != KD * 2 || kp->nb[2] != (size_t) hd * 2 || qp->nb[0] != 4 || qp->nb[2] != (size_t) hd * 4) continue;
        float scale; memcpy(&scale, scn->op_params, 4);
        if (scale != 1.0f / sqrtf((float) hd) || sm->src[0]->op_params[0] != (int32_t)(n_kv - 1)) continue;
        // ---- V.P -> PERMUTE -> CONT
        const int ikqv = users[i][0];
        const ggml_tensor * kqv = node(ikqv);
        if (kqv->op != GGML_OP_MUL_MAT || kqv->src[1] != sm || !only_local(ikqv, 1)) continue;
        const ggml_tensor * vv = kqv->src[0];
        if (vv->type != GGML_TYPE_F16 || vv->ne[0] != n_kv || vv->ne[1] != hd || vv->ne[2] != nkv || vv->ne[3] != 1 || vv->nb[0] != 2 || vv->nb[1] % 2) continue;
        const int64_t ML = (int64_t)(vv->nb[1] / 2);
        if (ML < n_kv || vv->nb[2] != (size_t) hd * vv->nb[1] || !cllm_attn_decode_supported((int) nh, (int) nkv, (int) hd, ML)) continue;
        const int iperm = users[ikqv][0];
        const ggml_tensor * perm = node(iperm);
        if (perm->op != GGML_OP_PERMUTE || perm->src[0] != kqv || !only_local(iperm, 1) || perm->ne[0] != hd || perm->ne[1] != nh || perm->ne[2] != 1) continue;
        const int icont = users[iperm][0];
        const ggml_tensor * cont = node(icont);
        if (cont->op != GGML_OP_CONT || cont->src[0] != perm || !f32_dense(cont) || !ggml_is_contiguous(cont) || ((uintptr_t) cont->data & 15)) continue;
        // ---- q: PERMUTE(ROPE(RESHAPE*(q vector), pos))
        const ggml_tensor * qrope = qp->src[0];
        const int iqp = find(qp), iqrope = find(qrope);
        if (qrope->op != GGML_OP_ROPE || iqp < 0 || iqrope < 0 || !only_local(iqp, 1) || !only_local(iqrope, 1)) continue;
        const ggml_tensor * pos = qrope->src[1];
        if (!pos || pos->type != GGML_TYPE_I32 || ggml_nelements(pos) != 1 || !pos->data) continue;
        // ---- the cache writes of this step: K row n_kv - 1 <- RESHAPE*(ROPE(RESHAPE*(k vector), pos)), by SET_ROWS(k_cache, ., pos)
        //      (KVCacheAttention) or by CPY into a view of that row (the sliding-window attention classes); CPY(TRANSPOSE(v vector) -> column n_kv - 1)
        int iset = -1, icpy = -1;
        const char * vcol = (const char *) vv->data + (size_t)(n_kv - 1) * 2, * krow = (const char *) kp->data + (size_t)(n_kv - 1) * KD * 2;
        for (int j = ikq - 1; j >= 0 && j >= ikq - 64 && (iset < 0 || icpy < 0); j--) {
            const ggml_tensor * t = node(j);
            if (iset < 0 && t->op == GGML_OP_SET_ROWS && t->data == kp->data) iset = j;
            if (iset < 0 && t->op == GGML_OP_CPY && t->src[1] && (const char *) t->src[1]->data == krow && (const char *) kp->data != (const char *) vv->data) iset = j;
            if (icpy < 0 && t->op == GGML_OP_CPY && t->src[1] && (const char *) t->src[1]->data == vcol && j != iset) icpy = j;
        }
        if (iset < 0 || icpy < 0 || iset == icpy) continue;
        const ggml_tensor * setr = node(iset), * cpy = node(icpy);
        if (setr->op == GGML_OP_SET_ROWS) {
            if (setr->type != GGML_TYPE_F16 || setr->ne[0] != (int64_t) KD || setr->nb[0] != 2 || setr->nb[1] != KD * 2 || !setr->src[1] || setr->src[1]->data != pos->data ||
                setr->src[1]->type != GGML_TYPE_I32 || ggml_nelements(setr->src[1]) != 1) continue;
        } else {
            const ggml_tensor * kdst = setr->src[1];
            if (kdst->type != GGML_TYPE_F16 || ggml_nelements(kdst) != (int64_t) KD || !ggml_is_contiguous(kdst) || setr->src[0]->type != GGML_TYPE_F32 ||
                ggml_nelements(setr->src[0]) != (int64_t) KD || !ggml_is_contiguous(setr->src[0])) continue;
        }
        const ggml_tensor * vdst = cpy->src[1], * vT = cpy->src[0];
        if (vdst->type != GGML_TYPE_F16 || vdst->ne[0] != 1 || vdst->ne[1] != (int64_t) KD || vdst->nb[1] != (size_t) ML * 2 || vT->type != GGML_TYPE_F32 ||
            vT->op != GGML_OP_TRANSPOSE || vT->ne[0] != 1 || vT->ne[1] != (int64_t) KD || vT->nb[1] != 4) continue;
        std::vector<int> kchain, qchain;
        const ggml_tensor * krope = through_reshapes(setr->src[0], kchain);
        if (!krope || krope->op != GGML_OP_ROPE || !krope->src[1] || krope->src[1]->data != pos->data || ggml_nelements(krope) != (int64_t) KD || krope->ne[0] != hd) continue;
        if (memcmp(krope->op_params + 1, qrope->op_params + 1, 10 * sizeof(int32_t)) || krope->src[2] != qrope->src[2]) continue;
        kchain.push_back(find(krope));
        const ggml_tensor * ks = through_reshapes(krope->src[0], kchain);
        const ggml_tensor * qs = through_reshapes(qrope->src[0], qchain);
        const ggml_tensor * vs = vT->src[0];
        const int iks = find(ks), iqs = find(qs), ivs = find(vs), ivT = find(vT);

        fused_attn A;
        A.pos = (const int32_t *) pos->data; A.nh = (int) nh; A.nkv = (int) nkv; A.hd = (int) hd; A.n_kv = n_kv; A.ML = ML;
        A.k_cache = kp->data; A.v_cache = vv->data; A.out = (float *) cont->data;
        A.mode = qrope->op_params[2]; memcpy(&A.freq_base, qrope->op_params + 5, 4);
        float freq_scale, ext_factor, attn_factor;
        memcpy(&freq_scale, qrope->op_params + 6, 4); memcpy(&ext_factor, qrope->op_params + 7, 4); memcpy(&attn_factor, qrope->op_params + 8, 4);
        bool l2 = max_level >= 2 && (A.mode == 0 || A.mode == 2) && qrope->op_params[1] == hd && freq_scale == 1.0f && ext_factor == 0.0f && attn_factor == 1.0f &&
                  !qrope->src[2] && hd % 4 == 0 && local[iset] == 0 && local[icpy] == 0 && only_local(ivT, 1);
        l2 = l2 && iks >= 0 && iqs >= 0 && ivs >= 0 && only_local(iks, 1) && only_local(iqs, 1) && only_local(ivs, 1) &&
             writer[iks] >= 0 && writer[iqs] >= 0 && writer[ivs] >= 0 && ggml_nelements(ks) == (int64_t) KD && ggml_nelements(vs) == (int64_t) KD && ggml_nelements(qs) == hd * nh;
        for (int j : kchain) l2 = l2 && only_local(j, 1);
        for (int j : qchain) l2 = l2 && only_local(j, 1);
        if (l2) {
            A.level = 2; A.wq = writer[iqs]; A.wk = writer[iks]; A.wv = writer[ivs];
            for (int j : kchain) P.skip[j] = 1;
            for (int j : qchain) P.skip[j] = 1;
            P.skip[iqrope] = P.skip[iset] = P.skip[icpy] = 1;
        } else {
            A.level = 1; A.q = (const float *) qp->data;
        }
        P.skip[ikq] = P.skip[i] = P.skip[ikqv] = 1;         // (SCALE and DIAG_MASK_INF are already swallowed by the soft_max pattern)
        P.attn[icont] = (int) P.attns.size(); P.attns.push_back(A);
    }
}

fuse_plan make_plan(ggml_cgraph * g) {
    const int n = ggml_graph_n_nodes(g);
    fuse_plan P; P.skip.assign(n, 0); P.mv.assign(n, -1); P.sm_src.assign(n, -1); P.attn.assign(n, -1); P.alt.assign(n, ALT_NONE); P.moe.assign(n, -1); P.pf.assign(n, -1);
    static const bool off = getenv("CLLM_HIP_NO_FUSE") != nullptr;
    if (off || n < 8) return P;
    std::vector<int> local(n, 0), writer(n, -1);        // writer[i]: entry of mvs whose launch produces node i
    static thread_local node_index find;
    static thread_local user_lists users;
    static thread_local std::vector<int> edges;
    find.build(g, n);
    // uses inside this graph (views count as uses of their source, exactly like ggml's use counts)
    edges.clear();
    for (int j = 0; j < n; j++) {
        const ggml_tensor * t = ggml_graph_node(g, j);
        for (int k = 0; k < GGML_MAX_SRC; k++) if (t->src[k]) { const int i = find(t->src[k]); if (i >= 0 && i < j) { local[i]++; edges.push_back(i); edges.push_back(j); } }
    }
    users.start.assign(n + 1, 0);
    for (int i = 0; i < n; i++) users.start[i + 1] = users.start[i] + local[i];
    users.list.resize(edges.size() / 2);
    {
        std::vector<int> fill(users.start.begin(), users.start.end() - 1);
        for (size_t e = 0; e < edges.size(); e += 2) users.list[fill[edges[e]]++] = edges[e + 1];      // edges are in consumer order: so is every list
    }
    auto only_local = [&](int i, int uses) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        return !(t->flags & GGML_TENSOR_FLAG_OUTPUT) && local[i] == uses && ggml_node_get_use_count(g, i) == uses;
    };
    auto node_of = [&](const ggml_tensor * t, int before) { for (int i = before - 1; i >= 0 && i >= before - 64; i--) if (ggml_graph_node(g, i) == t) return i; return -1; };
    // a single-column quantized MUL_MAT the decode mat-vec takes; row length: 16384 with the norm prologue, 32768 with the others (gemv_decode.hip)
    auto mv_ok = [&](const ggml_tensor * mm, int64_t kmax = 32768) {
        const ggml_tensor * w = mm->src[0], * x = mm->src[1];
        return mm->op == GGML_OP_MUL_MAT && is_q(w->type) && w->ne[2] == 1 && w->ne[3] == 1 && w->nb[1] == ggml_row_size(w->type, w->ne[0]) && ((uintptr_t) w->data & 15) == 0 &&
               x->type == GGML_TYPE_F32 && x->ne[1] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && f32_vec(mm) &&
               w->ne[0] <= kmax && (uint64_t) w->ne[1] * w->nb[1] < (1ull << 32);
    };
    for (int i = 0; i < n; i++) {
        ggml_tensor * t = ggml_graph_node(g, i);
        // ---- RMS_NORM -> MUL(weight) -> mat-vecs
        if (t->op == GGML_OP_MUL && t->src[0] && t->src[0]->op == GGML_OP_RMS_NORM && f32_vec(t) && f32_vec(t->src[0]->src[0]) && f32_vec(t->src[1]) &&
            t->src[1]->ne[0] == t->ne[0] && t->ne[0] % 256 == 0) {
            const int r = node_of(t->src[0], i);
            bool ok = r >= 0 && only_local(r, 1) && only_local(i, local[i]) && local[i] > 0;
            for (int j : users[i]) { const ggml_tensor * c = ggml_graph_node(g, j); ok = ok && mv_ok(c, 16384) && c->src[1] == t; }
            if (ok) {
                float eps; memcpy(&eps, t->src[0]->op_params, 4);
                for (int j : users[i]) {
                    fused_mv f; f.pro = 1; f.px = (const float *) t->src[0]->src[0]->data; f.pw = (const float *) t->src[1]->data; f.eps = eps; f.dst = (float *) ggml_graph_node(g, j)->data; f.node = j;
                    P.mv[j] = writer[j] = (int) P.mvs.size(); P.mvs.push_back(f);
                }
                P.skip[r] = P.skip[i] = 1;
            }
        }
        // ---- UNARY(SILU) -> MUL(up) -> mat-vec
        if (t->op == GGML_OP_MUL && t->src[0] && t->src[0]->op == GGML_OP_UNARY && ggml_get_unary_op(t->src[0]) == GGML_UNARY_OP_SILU && f32_vec(t) && f32_vec(t->src[0]->src[0]) &&
            f32_vec(t->src[1]) && t->src[1]->ne[0] == t->ne[0] && local[i] == 1) {
            const int u = node_of(t->src[0], i), j = users[i][0];
            const ggml_tensor * c = ggml_graph_node(g, j);
            if (u >= 0 && only_local(u, 1) && only_local(i, 1) && mv_ok(c) && c->src[1] == t && t->ne[0] % 8 == 0 && c->src[0]->ne[0] <= 32768) {
                fused_mv f; f.pro = 4; f.px = (const float *) t->src[0]->src[0]->data; f.pw = (const float *) t->src[1]->data; f.dst = (float *) c->data;
                f.node = j; f.ga = find(t->src[0]->src[0]); f.gb = find(t->src[1]);
                P.mv[j] = writer[j] = (int) P.mvs.size(); P.mvs.push_back(f);
                P.skip[u] = P.skip[i] = 1;
            }
        }
        // ---- SCALE -> DIAG_MASK_INF -> SOFT_MAX (no mask tensor, no ALiBi)
        if (t->op == GGML_OP_SOFT_MAX && !t->src[1] && t->src[0]->op == GGML_OP_DIAG_MASK_INF && t->src[0]->src[0]->op == GGML_OP_SCALE) {
            const int dm = node_of(t->src[0], i), sc = dm >= 0 ? node_of(t->src[0]->src[0], dm) : -1;
            float s1, mb, bias; memcpy(&s1, t->op_params, 4); memcpy(&mb, (const float *) t->op_params + 1, 4);
            if (sc >= 0) memcpy(&bias, (const float *) ggml_graph_node(g, sc)->op_params + 1, 4);
            if (sc >= 0 && only_local(dm, 1) && only_local(sc, 1) && s1 == 1.0f && mb == 0.0f && bias == 0.0f) { P.sm_src[i] = sc; P.skip[dm] = P.skip[sc] = 1; }
        }
    }
    // ---- mat-vec -> ADD: after the prologue patterns, so that it composes with them
    for (int i = 0; i < n; i++) {
        ggml_tensor * a = ggml_graph_node(g, i);
        if (a->op != GGML_OP_ADD || !f32_vec(a)) continue;
        for (int side = 0; side < 2; side++) {
            const ggml_tensor * mm = a->src[side], * r = a->src[1 - side];
            if (!mm || mm->op != GGML_OP_MUL_MAT || !mv_ok(mm) || !f32_vec(r) || r->ne[0] != a->ne[0]) continue;
            const int j = node_of(mm, i);
            if (j < 0 || !only_local(j, 1) || P.skip[j]) continue;
            if (P.mv[j] < 0) { fused_mv f; f.pro = 2; f.px = (const float *) mm->src[1]->data; f.node = j; P.mv[j] = (int) P.mvs.size(); P.mvs.push_back(f); }
            fused_mv & f = P.mvs[P.mv[j]];
            const size_t kbytes = (size_t) mm->src[0]->ne[0] * 4, nbytes = (size_t) a->ne[0] * 4;
            if (overlap(a->data, nbytes, f.px, kbytes) || (f.pw && overlap(a->data, nbytes, f.pw, kbytes))) { if (f.pro == 2 && !f.resid && !f.dst) { P.mvs.pop_back(); P.mv[j] = -1; } continue; }
            f.resid = (const float *) r->data; f.resid_t = r; f.dst = (float *) a->data;
            P.skip[i] = 1;
            writer[j] = -1; writer[i] = P.mv[j];
            break;
        }
    }
    // ---- prefill: the same three patterns around a many-column quantized MUL_MAT (cllm_op_mul_mat_ex; the runner's prefill has them too)
    static const bool no_pf = getenv("CLLM_HIP_NO_PREFILL_FUSE") != nullptr;
    static const int pf_min = cllm_mul_mat_ex_min_cols();
    auto pf_ok = [&](const ggml_tensor * mm) {
        if (mm->op != GGML_OP_MUL_MAT) return false;
        const ggml_tensor * w = mm->src[0], * x = mm->src[1];
        const uintptr_t wal = w->type == GGML_TYPE_Q4_K ? 15 : w->type == GGML_TYPE_Q4_1 ? 3 : 1;
        return is_q(w->type) && w->ne[2] == 1 && w->ne[3] == 1 && w->nb[1] == ggml_row_size(w->type, w->ne[0]) && !((uintptr_t) w->data & wal) && !(w->nb[1] & wal) &&
               x->type == GGML_TYPE_F32 && x->ne[2] == 1 && x->ne[3] == 1 && x->ne[1] >= pf_min && x->ne[1] <= 65535 && x->nb[0] == 4 && x->nb[1] % 16 == 0 && !((uintptr_t) x->data & 15) &&
               mm->type == GGML_TYPE_F32 && mm->nb[0] == 4 && mm->nb[1] % 4 == 0 && mm->ne[2] == 1 && mm->ne[3] == 1;
    };
    auto act_class = [](ggml_type t) { return t == GGML_TYPE_Q4_K ? 0 : t == GGML_TYPE_Q4_1 ? 1 : 2; };
    auto uses_wdata = [](const ggml_tensor * t) { return t->op == GGML_OP_MUL_MAT || t->op == GGML_OP_MUL_MAT_ID || t->op == GGML_OP_FLASH_ATTN_EXT; };
    if (!no_pf) for (int i = 0; i < n; i++) {
        ggml_tensor * t = ggml_graph_node(g, i);
        if (P.skip[i] || t->op != GGML_OP_MUL || t->type != GGML_TYPE_F32 || t->ne[1] < pf_min || t->ne[2] != 1 || t->ne[3] != 1 || !ggml_is_contiguous(t)) continue;
        const ggml_tensor * a = t->src[0], * b = t->src[1];
        // RMS_NORM -> MUL(weight vector) -> projections
        const int ir = find(a);
        if (a->op == GGML_OP_RMS_NORM && ir >= 0 && !P.skip[ir] && only_local(ir, 1) && only_local(i, local[i]) && local[i] > 0 && a->src[0]->type == GGML_TYPE_F32 &&
            ggml_is_contiguous(a->src[0]) && !((uintptr_t) a->src[0]->data & 15) && t->ne[0] % 4 == 0 && t->ne[0] <= 16384 &&
            b->type == GGML_TYPE_F32 && ggml_is_contiguous(b) && ggml_nelements(b) == t->ne[0] && !((uintptr_t) b->data & 15)) {
            bool ok = true;
            for (int j : users[i]) { const ggml_tensor * c = ggml_graph_node(g, j); ok = ok && pf_ok(c) && c->src[1] == t && P.pf[j] < 0 && !P.skip[j]; }
            if (!ok) continue;
            int prev = -1;
            for (int j : users[i]) {
                pf_mm F; F.pro = 1; F.x = a->src[0]; F.w2 = b; memcpy(&F.eps, a->op_params, 4);
                if (prev >= 0 && act_class(ggml_graph_node(g, j)->src[0]->type) == act_class(ggml_graph_node(g, prev)->src[0]->type)) {
                    bool clean = true;                 // nothing between the two projections may touch the module's act scratch
                    for (int k = prev + 1; k < j; k++) clean = clean && !uses_wdata(ggml_graph_node(g, k));
                    if (clean) F.pro = 5;
                }
                P.pf[j] = (int) P.pfs.size(); P.pfs.push_back(F);
                prev = j;
            }
            P.skip[ir] = P.skip[i] = 1;
            continue;
        }
        // UNARY(SILU)(gate) -> MUL(up) -> MUL_MAT
        for (int side = 0; side < 2; side++) {
            const ggml_tensor * u = side ? b : a, * o = side ? a : b;
            const int iu = find(u);
            if (u->op != GGML_OP_UNARY || ggml_get_unary_op(u) != GGML_UNARY_OP_SILU || iu < 0 || P.skip[iu] || !only_local(iu, 1) || !only_local(i, 1)) continue;
            const int j = users[i][0];
            const ggml_tensor * c = ggml_graph_node(g, j), * gate = u->src[0];
            if (!pf_ok(c) || c->src[1] != t || P.pf[j] >= 0 || P.skip[j] || !ggml_are_same_shape(gate, t) || !ggml_are_same_shape(o, t) || gate->type != GGML_TYPE_F32 || o->type != GGML_TYPE_F32 ||
                !ggml_is_contiguous(gate) || !ggml_is_contiguous(o) || ((uintptr_t) gate->data & 15) || ((uintptr_t) o->data & 15) || t->ne[0] % 4) continue;
            pf_mm F; F.pro = 4; F.x = gate; F.w2 = o;
            P.pf[j] = (int) P.pfs.size(); P.pfs.push_back(F);
            P.skip[iu] = P.skip[i] = 1;
            break;
        }
    }
    // MUL_MAT -> ADD(residual): the ADD directly follows (only views in between: its destination is then not the live memory of anything that still runs)
    if (!no_pf) for (int i = 0; i < n; i++) {
        ggml_tensor * ad = ggml_graph_node(g, i);
        if (P.skip[i] || ad->op != GGML_OP_ADD || ad->type != GGML_TYPE_F32 || !ggml_is_contiguous(ad)) continue;
        for (int side = 0; side < 2; side++) {
            const ggml_tensor * mm = ad->src[side], * r = ad->src[1 - side];
            const int j = mm ? find(mm) : -1;
            if (j < 0 || j >= i || !pf_ok(mm) || P.skip[j] || !only_local(j, 1) || !r || r->type != GGML_TYPE_F32 || !ggml_are_same_shape(r, ad) || !ggml_are_same_shape(mm, ad) || !ggml_is_contiguous(r) ||
                !ggml_is_contiguous(mm)) continue;
            bool adjacent = true;
            for (int k = j + 1; k < i; k++) { const ggml_op op = ggml_graph_node(g, k)->op; adjacent = adjacent && (op == GGML_OP_RESHAPE || op == GGML_OP_VIEW || op == GGML_OP_PERMUTE || op == GGML_OP_TRANSPOSE); }
            if (!adjacent || (ad->data != r->data && overlap(ad->data, ggml_nbytes(ad), r->data, ggml_nbytes(r)))) continue;
            if (P.pf[j] < 0) { pf_mm F; F.pro = 0; F.x = mm->src[1]; P.pf[j] = (int) P.pfs.size(); P.pfs.push_back(F); }
            P.pfs[P.pf[j]].resid = r; P.pfs[P.pf[j]].out = ad;
            P.skip[i] = 1;
            break;
        }
    }
    plan_attention(g, P, local, writer, users, find);
    // ---- merged launches over packed weights (the decision whether a packed copy exists is taken at run time: get_pack)
    auto is_bias = [&](const fused_mv & f) {     // the epilogue operand is a weight vector of the projection's length: never an activation
        const ggml_tensor * w = ggml_graph_node(g, f.node)->src[0];
        return f.resid_t && f.resid_t->op == GGML_OP_NONE && ours(f.resid_t) && f.resid_t->type == GGML_TYPE_F32 && ggml_is_contiguous(f.resid_t) &&
               ggml_nelements(f.resid_t) == w->ne[1] && ((uintptr_t) f.resid_t->data & 15) == 0;
    };
    auto same_input = [&](const fused_mv & a, const fused_mv & b, bool with_bias) {
        const ggml_tensor * wa = ggml_graph_node(g, a.node)->src[0], * wb = ggml_graph_node(g, b.node)->src[0];
        const bool epi_ok = with_bias ? (is_bias(a) && is_bias(b)) : (!a.resid && !b.resid);
        return a.pro == 1 && b.pro == 1 && a.px == b.px && a.pw == b.pw && a.eps == b.eps && epi_ok && a.group < 0 && b.group < 0 && !a.alias && !b.alias &&
               wa->type == wb->type && wa->ne[0] == wb->ne[0] && wa->nb[1] == wb->nb[1];
    };
    for (const fused_attn & A : P.attns) {
        if (A.level != 2 || A.wq == A.wk || A.wq == A.wv || A.wk == A.wv) continue;
        const bool with_bias = P.mvs[A.wq].resid != nullptr;
        if (!same_input(P.mvs[A.wq], P.mvs[A.wk], with_bias) || !same_input(P.mvs[A.wq], P.mvs[A.wv], with_bias)) continue;
        merge_group G; G.n = 3; G.member[0] = A.wq; G.member[1] = A.wk; G.member[2] = A.wv; G.bias = with_bias;
        for (int k = 0; k < 3; k++) P.mvs[G.member[k]].group = (int) P.groups.size();
        P.groups.push_back(G);
    }
    for (size_t m = 0; m < P.mvs.size(); m++) {
        const fused_mv & d = P.mvs[m];
        if (d.pro != 4 || d.ga < 0 || d.gb < 0 || d.ga == d.gb || P.mv[d.ga] < 0 || P.mv[d.gb] < 0 || P.skip[d.ga] || P.skip[d.gb]) continue;
        const fused_mv & a = P.mvs[P.mv[d.ga]], & b = P.mvs[P.mv[d.gb]];
        const ggml_tensor * ta = ggml_graph_node(g, d.ga), * tb = ggml_graph_node(g, d.gb);
        if (!same_input(a, b, false) || a.dst != (float *) ta->data || b.dst != (float *) tb->data || !only_local(d.ga, 1) || !only_local(d.gb, 1)) continue;
        const int64_t F = ta->ne[0];
        if (tb->ne[0] != F || F % 8 || ta->src[0]->ne[1] != F || tb->src[0]->ne[1] != F) continue;
        merge_group G; G.n = 2; G.member[0] = P.mv[d.ga]; G.member[1] = P.mv[d.gb]; G.interleave = true; G.consumer = (int) m;
        P.mvs[G.member[0]].group = P.mvs[G.member[1]].group = (int) P.groups.size();
        P.groups.push_back(G);
    }
    for (int j = 0; j < n; j++) if (P.mv[j] >= 0 && !P.mvs[P.mv[j]].dst) P.mvs[P.mv[j]].dst = (float *) ggml_graph_node(g, j)->data;
    // A fused mat-vec has no grid-wide barrier between its prologue (EVERY workgroup reads all of px / pw) and the first stores of dst.
    // ggml-alloc allocates a node before it frees the node's parents, but the tensors a fused launch swallows (the norm's input is not one of
    // them; `up` of SILU->MUL(up)->MUL_MAT, the normalised activation ...) may already be free when dst is placed -- so dst can land on an
    // input.  Such launches write to module scratch and are copied into place afterwards (graph_compute); an in-place residual (dst == resid)
    // is fine: every element is read and written by the same lane.
    static const bool force_stage = getenv("CLLM_HIP_FORCE_STAGE") != nullptr;      // (tests: take the staged path everywhere)
    for (fused_mv & f : P.mvs) {
        if (f.node < 0 || !f.dst) continue;
        const ggml_tensor * w = ggml_graph_node(g, f.node)->src[0];
        const size_t nb = (size_t) w->ne[1] * 4, kb = (size_t) w->ne[0] * 4;
        f.alias = force_stage || (f.px && overlap(f.dst, nb, f.px, kb)) || (f.pw && overlap(f.dst, nb, f.pw, kb));
    }
    for (fused_attn & A : P.attns) if (A.level == 1) A.alias = force_stage || overlap(A.out, (size_t) A.nh * A.hd * 4, A.q, (size_t) A.nh * A.hd * 4);
    // ---- what the patterns above left: element-wise pairs and the tail of a sparse-MoE block
    auto strip = [&](const ggml_tensor * t) { while (t && t->op == GGML_OP_RESHAPE) t = t->src[0]; return t; };
    auto all_once = [&](const ggml_tensor * from, const ggml_tensor * to) {      // the RESHAPE chain from `from` down to (excluding) `to`: every node used once
        for (const ggml_tensor * t = from; t && t != to; t = t->src[0]) { if (t->op != GGML_OP_RESHAPE || !only_local(find(t), 1)) return false; }
        return true;
    };
    for (int i = 0; i < n; i++) {
        ggml_tensor * t = ggml_graph_node(g, i);
        if (P.skip[i] || t->op != GGML_OP_MUL || t->type != GGML_TYPE_F32) continue;
        const ggml_tensor * a = t->src[0], * b = t->src[1];
        // UNARY(SILU) on either side, same shape, dense
        for (int side = 0; side < 2; side++) {
            const ggml_tensor * u = side ? b : a, * o = side ? a : b;
            const int iu = find(u);
            if (u->op == GGML_OP_UNARY && ggml_get_unary_op(u) == GGML_UNARY_OP_SILU && iu >= 0 && !P.skip[iu] && only_local(iu, 1) && ggml_are_same_shape(u, o) &&
                ggml_are_same_shape(u, t) && f32_dense(u->src[0]) && f32_dense(o) && f32_dense(t) && ggml_is_contiguous(u->src[0]) && ggml_is_contiguous(o) && ggml_is_contiguous(t)) {
                P.alt[i] = side ? ALT_MUL_SILU : ALT_SILU_MUL; P.skip[iu] = 1;
                // both operands straight from one-token MUL_MAT_IDs over the same activation and ids: candidate for the merged expert launch
                const ggml_tensor * gm = u->src[0], * um = o;
                const int igm = find(gm), ium = find(um);
                if (gm->op == GGML_OP_MUL_MAT_ID && um->op == GGML_OP_MUL_MAT_ID && igm >= 0 && ium >= 0 && only_local(igm, 1) && only_local(ium, 1) &&
                    gm->src[1] == um->src[1] && gm->src[2] == um->src[2] && gm->src[2]->ne[1] == 1 && gm->src[2]->nb[0] == 4 &&
                    gm->src[0]->type == um->src[0]->type && is_q(gm->src[0]->type) && ggml_are_same_shape(gm->src[0], um->src[0]) && gm->src[0]->nb[1] == um->src[0]->nb[1] &&
                    gm->src[0]->ne[1] % 8 == 0 && gm->src[0]->ne[0] <= 32768 && gm->src[1]->nb[0] == 4 && gm->src[1]->nb[1] % 16 == 0 && ((uintptr_t) gm->src[1]->data & 15) == 0) {
                    moe_gate_up G; G.mul = i; G.gate = igm; G.up = ium; G.unary = iu;
                    P.gus.push_back(G);
                }
                break;
            }
        }
        if (P.alt[i]) continue;
        // RMS_NORM -> MUL(weight vector)
        const int ir = find(a);
        if (a->op == GGML_OP_RMS_NORM && ir >= 0 && !P.skip[ir] && only_local(ir, 1) && f32_dense(a->src[0]) && ggml_is_contiguous(a->src[0]) && ggml_is_contiguous(t) &&
            b->type == GGML_TYPE_F32 && ggml_is_contiguous(b) && ggml_nelements(b) == t->ne[0] && b->ne[0] == t->ne[0]) {
            // one token, and among the consumers a small router mat-vec -> SOFT_MAX -> TOP_K: the whole head of the sparse-MoE block in one launch
            int imm = -1, ism = -1, itk = -1;
            if (ggml_nelements(t) == t->ne[0] && t->ne[0] % 4 == 0 && !(((uintptr_t) t->data | (uintptr_t) a->src[0]->data | (uintptr_t) b->data) & 15) &&
                (t->data == a->src[0]->data || !overlap(t->data, (size_t) t->ne[0] * 4, a->src[0]->data, (size_t) t->ne[0] * 4)))
                for (int u : users[i]) {
                    const ggml_tensor * c = ggml_graph_node(g, u);
                    if (c->op == GGML_OP_MUL_MAT && c->src[1] == t && mv_ok(c, 16384) && c->src[0]->ne[1] <= 64 && P.mv[u] < 0 && !P.skip[u] && only_local(u, 1)) { imm = u; break; }
                }
            if (imm >= 0) {
                const int u = users[imm][0];
                const ggml_tensor * sm = ggml_graph_node(g, u);
                float s1, mb; memcpy(&s1, sm->op_params, 4); memcpy(&mb, (const float *) sm->op_params + 1, 4);
                if (sm->op == GGML_OP_SOFT_MAX && !sm->src[1] && sm->src[0] == ggml_graph_node(g, imm) && s1 == 1.0f && mb == 0.0f && f32_vec(sm) && !P.skip[u] && P.sm_src[u] < 0 &&
                    !(sm->flags & GGML_TENSOR_FLAG_OUTPUT)) ism = u;
            }
            if (ism >= 0) {
                int cnt = 0;
                for (int u : users[ism]) { const ggml_tensor * c = ggml_graph_node(g, u); if (c->op == GGML_OP_TOP_K && c->src[0] == ggml_graph_node(g, ism)) { itk = u; cnt++; } }
                const ggml_tensor * tk = itk >= 0 ? ggml_graph_node(g, itk) : nullptr;
                if (cnt != 1 || tk->type != GGML_TYPE_I32 || tk->nb[0] != 4 || ggml_nelements(tk) != tk->ne[0] || tk->ne[0] > ggml_graph_node(g, ism)->ne[0] || P.skip[itk]) itk = -1;
            }
            if (itk >= 0) {
                // everything else that reads the normalised activation or the probabilities must run after the launch (which stands at the TOP_K node)
                std::function<bool(int)> later = [&](int j) {
                    for (int u : users[j]) {
                        if (u == imm || u == ism || u == itk) continue;
                        const ggml_tensor * c = ggml_graph_node(g, u);
                        const bool noop = c->op == GGML_OP_RESHAPE || c->op == GGML_OP_VIEW || c->op == GGML_OP_PERMUTE || c->op == GGML_OP_TRANSPOSE;
                        if (noop ? !later(u) : u < itk) return false;
                    }
                    return true;
                };
                if (later(i) && later(ism)) {
                    moe_router R; R.x = a->src[0]; R.w = b; R.gate = ggml_graph_node(g, imm)->src[0]; R.xnorm = t; R.probs = ggml_graph_node(g, ism);
                    memcpy(&R.eps, a->op_params, 4);
                    R.itk = itk;
                    std::function<void(int, std::vector<int> &)> ends = [&](int j, std::vector<int> & out) {      // the end consumers of node j behind no-op views
                        for (int u : users[j]) {
                            if (u == imm || u == ism || u == itk) continue;
                            const ggml_tensor * c = ggml_graph_node(g, u);
                            if (c->op == GGML_OP_RESHAPE || c->op == GGML_OP_VIEW || c->op == GGML_OP_PERMUTE || c->op == GGML_OP_TRANSPOSE) ends(u, out); else out.push_back(u);
                        }
                    };
                    ends(i, R.xn_users); ends(ism, R.out_users); ends(itk, R.out_users);
                    P.skip[ir] = P.skip[i] = P.skip[imm] = P.skip[ism] = 1;
                    P.alt[itk] = ALT_MOE_ROUTER; P.moe[itk] = (int) P.routers.size(); P.routers.push_back(R);
                    continue;
                }
            }
            P.alt[i] = ALT_RMS_NORM_MUL; P.skip[ir] = 1;
            continue;
        }
        // experts [H, k, T] * weights [1, k, T], weights = RESHAPE(DIV(RESHAPE(GET_ROWS(RESHAPE(probs), ids)), SUM_ROWS(same)))
        if (a->ne[3] != 1 || b->ne[0] != 1 || b->ne[1] != a->ne[1] || b->ne[2] != a->ne[2] || !f32_dense(a) || !ggml_is_contiguous(a) || !ggml_is_contiguous(t)) continue;
        const int64_t H = a->ne[0], k = a->ne[1], T = a->ne[2];
        const ggml_tensor * dv = strip(b);
        if (!dv || dv->op != GGML_OP_DIV || !all_once(b, dv) || !only_local(find(dv), 1)) continue;
        const ggml_tensor * wr = dv->src[0], * sr = dv->src[1];
        const int iwr = find(wr), isr = find(sr);
        if (sr->op != GGML_OP_SUM_ROWS || sr->src[0] != wr || !only_local(isr, 1) || wr->op != GGML_OP_RESHAPE || iwr < 0 || local[iwr] != 2 || ggml_node_get_use_count(g, iwr) != 2 ||
            wr->ne[0] != k || wr->ne[1] != T) continue;
        const ggml_tensor * gr = wr->src[0];
        const int igr = find(gr);
        if (gr->op != GGML_OP_GET_ROWS || !only_local(igr, 1) || gr->ne[0] != 1) continue;
        const ggml_tensor * probs = strip(gr->src[0]), * ids = gr->src[1];
        if (!probs || probs->type != GGML_TYPE_F32 || !ggml_is_contiguous(probs) || probs->ne[1] != T || ggml_nelements(probs) != probs->ne[0] * T ||
            ids->type != GGML_TYPE_I32 || ids->ne[0] != k || ids->ne[1] != T || ids->nb[0] != 4) continue;
        // the k slot views and the ADD chain
        if (local[i] != (int) k || ggml_node_get_use_count(g, i) != (int) k || (t->flags & GGML_TENSOR_FLAG_OUTPUT) || k < 1 || k > 64) continue;
        std::vector<int> views((size_t) k, -1);
        bool ok = true;
        for (int u : users[i]) {
            const ggml_tensor * v = ggml_graph_node(g, u);
            const size_t off = (size_t)((const char *) v->data - (const char *) t->data);
            if (v->op != GGML_OP_VIEW || v->src[0] != t || v->ne[0] != H || v->ne[1] != T || v->ne[2] != 1 || v->nb[1] != t->nb[2] || off % t->nb[1] || off / t->nb[1] >= (size_t) k ||
                views[off / t->nb[1]] >= 0 || !only_local(u, 1)) { ok = false; break; }
            views[off / t->nb[1]] = u;
        }
        if (!ok) continue;
        int cur = views[0];
        std::vector<int> adds;
        for (int64_t j = 1; j < k && ok; j++) {
            const int ia = users[cur][0];
            const ggml_tensor * ad = ggml_graph_node(g, ia);
            ok = ad->op == GGML_OP_ADD && ad->src[0] == ggml_graph_node(g, cur) && ad->src[1] == ggml_graph_node(g, views[(size_t) j]) && users[views[(size_t) j]][0] == ia &&
                 f32_dense(ad) && ggml_is_contiguous(ad) && (j == k - 1 || only_local(ia, 1));
            adds.push_back(ia); cur = ia;
        }
        if (!ok || k < 2) continue;
        fused_moe M; M.experts = a; M.probs = probs; M.ids = ids;
        int fin = cur;
        // every thread of k_moe_combine reads its element of all k slots (and of the residual), then writes its element of dst: dst may BE slot 0 of
        // a one-token experts tensor or the residual (ggml-alloc computes these ADDs in place), but must not overlap them any other way
        auto dst_ok = [&](const ggml_tensor * out, const ggml_tensor * resid) {
            const char * dp = (const char *) out->data;
            const size_t nb = (size_t)(H * T) * 4;
            if (overlap(dp, nb, a->data, ggml_nbytes(a)) && !(T == 1 && dp == (const char *) a->data)) return false;
            return !resid || dp == (const char *) resid->data || !overlap(dp, nb, resid->data, nb);
        };
        if (only_local(cur, 1)) {             // ... -> ADD(moe_out, residual)
            const int ia = users[cur][0];
            const ggml_tensor * ad = ggml_graph_node(g, ia);
            if (ad->op == GGML_OP_ADD && !P.skip[ia] && ad->src[0] == ggml_graph_node(g, cur) && f32_dense(ad->src[1]) && ggml_is_contiguous(ad->src[1]) &&
                ad->src[1]->ne[0] == H && ggml_nelements(ad->src[1]) == H * T && ggml_is_contiguous(ad) && dst_ok(ad, ad->src[1])) { M.resid = ad->src[1]; fin = ia; }
        }
        if (!M.resid && !dst_ok(ggml_graph_node(g, cur), nullptr)) continue;
        {   // the experts are the down projection's MUL_MAT_ID of one token over two slots: mat-vecs and tail in one launch (cllm_op_mul_mat_id_combine)
            const int ia = find(a);
            const ggml_tensor * out = ggml_graph_node(g, fin), * w = a->src[0], * x = a->src[1];
            static const bool no_down = getenv("CLLM_HIP_NO_MOE_DOWN_FUSE") != nullptr;
            if (!no_down && a->op == GGML_OP_MUL_MAT_ID && ia >= 0 && !P.skip[ia] && only_local(ia, 1) && k == 2 && T == 1 && a->src[2] == ids && is_q(w->type) && w->ne[3] == 1 &&
                w->nb[1] == ggml_row_size(w->type, w->ne[0]) && w->nb[2] % 16 == 0 && ((uintptr_t) w->data & 15) == 0 && w->ne[0] <= 32768 && (uint64_t) w->ne[1] * w->nb[1] < (1ull << 32) &&
                x->type == GGML_TYPE_F32 && x->ne[1] == 2 && x->ne[2] == 1 && x->ne[3] == 1 && x->nb[0] == 4 && x->nb[1] % 16 == 0 && ((uintptr_t) x->data & 15) == 0 &&
                probs->ne[0] == w->ne[2] && ggml_nelements(probs) == probs->ne[0] &&
                !overlap(out->data, (size_t) H * 4, x->data, ggml_nbytes(x)) && !overlap(out->data, (size_t) H * 4, probs->data, ggml_nbytes(probs)) &&
                !overlap(out->data, (size_t) H * 4, ids->data, ggml_nbytes(ids))) { M.down = ia; P.skip[ia] = 1; }
        }
        for (const ggml_tensor * r = b; r != dv; r = r->src[0]) P.skip[find(r)] = 1;
        P.skip[find(dv)] = P.skip[isr] = P.skip[iwr] = P.skip[igr] = 1;
        for (const ggml_tensor * r = gr->src[0]; r != probs; r = r->src[0]) if (find(r) >= 0) P.skip[find(r)] = 1;       // (RESHAPEs: no launches anyway)
        P.skip[i] = 1;
        for (int ia : adds) P.skip[ia] = 1;
        if (M.resid) P.skip[cur] = 1;
        P.skip[fin] = 0; P.alt[fin] = ALT_MOE_COMBINE; P.moe[fin] = (int) P.moes.size(); P.moes.push_back(M);
    }
    return P;
}

// ---- launch-list replay ---------------------------------------------------------------------------------------------------
// The reference rebuilds, re-splits and re-allocates its graph for every token (src/models.cpp:1244-1312), but for a decode step the
// result is the same list of C-ABI calls with the same arguments as the token before: everything that changes (token id, position)
// is read from device memory by the kernels, and ggml's allocator hands out the same addresses.  graph_compute therefore walks the
// nodes twice: first only serializing every call's arguments; if the bytes equal those of a captured graph, that graph is launched
// (one host call, no per-launch gaps on the GPU) instead of the ~160 launches; if they equal the previous token's list, this token's
// launches are captured on the way.  Any difference -- another address, shape, scalar, fusion decision -- simply issues the calls.
// Measured (Llama-3-8B Q4_K through the unmodified host, one MI355X): the host time inside graph_compute drops from 586 to 74 us per
// token, but the token is GPU-bound -- issue + synchronize is 1.75 ms either way (the launches were already running ahead of the GPU) --
// so wall time barely moves -- but the host thread is free ~0.5 ms earlier per token and the GPU sees no launch gaps.  ON by default
// (CLLM_HIP_GRAPH=0 turns it off); every replayed list is byte-identical to the calls it stands for (tests/test_gpu_dropin.py).
struct sig_writer {
    std::vector<uint8_t> & b;
    void raw(const void * p, size_t n) { const uint8_t * q = (const uint8_t *) p; b.insert(b.end(), q, q + n); }
    template <class T> void put(const T & v) { static_assert(std::is_trivially_copyable<T>::value, "pod"); raw(&v, sizeof(v)); }
    void arg(const cllm_tensor * t) { if (!t) { put((uint8_t) 0); return; } put((uint8_t) 1); put(t->type); raw(t->ne, sizeof(t->ne)); raw(t->nb, sizeof(t->nb)); put(t->data); }
    void arg(cllm_tensor * t) { arg((const cllm_tensor *) t); }
    void arg(const cllm_rope_params * r) { if (r) put(*r); else put((uint8_t) 0); }
    void arg(cllm_rope_params * r) { arg((const cllm_rope_params *) r); }
    void arg(std::nullptr_t) { put((uint8_t) 0); }
    template <class T> typename std::enable_if<std::is_arithmetic<T>::value || std::is_pointer<T>::value || std::is_enum<T>::value>::type arg(T v) { put(v); }
    std::vector<size_t> * marks = nullptr;        // (CLLM_HIP_SIG_DEBUG: where every call starts)
    template <class... A> void call(const void * fn, A... a) { if (marks) marks->push_back(b.size()); put(fn); int dummy[] = { 0, (arg(a), 0)... }; (void) dummy; }
};

// ---- decode-ahead.  chatllm's host is synchronous: compute, synchronize, read the logits, sample, build the next graph (~0.5 ms for Llama-3-8B),
//      compute ...  -- the GPU idles while the host works.  In the steady state of a greedy decode the next step is known before the host asks for it:
//      same captured launch list, token = argmax(logits), every position + 1.  So right after the host has read the logits the module starts that step
//      itself (argmax + scalar updates on the device, then the captured graph); when the host's graph_compute arrives, its launch list and the scalars
//      it wrote are compared with the prediction: equal -> the work is already running (or done), nothing is launched; different -> the step runs again
//      the normal way (what ran ahead only touched the logits, scratch and the cache row of the next position, all of which the real step rewrites).
//      Guards: only after a replayed step; only if the logits are the graph's single output; positions stay inside the caches; any other access through
//      the buffer interface waits for the step running ahead (ahead_quiesce) and the logits stay readable from a snapshot; two misses in a row switch
//      it off for the next 64 graphs (a sampling host costs itself at most a few % that way).  CLLM_HIP_AHEAD=0 turns it off.
// chained (round 6): called from graph_compute itself, right behind the launch of the step the host asked for -- the snapshot of that step's outputs, the arg-max and the next
// step are queued on the stream before the host has even synchronized; synchronize() then waits for `ev` (outputs snapshotted, next token known) instead of the whole stream, and
// the GPU goes from one step into the next without waiting for the host to wake up, read and come back (~80 us per token: 682 vs 722 tok/s through the host in round 5).
// Still at most ONE step runs ahead of what the host has asked for.
void ahead_finish_event(hip_backend_ctx * c) {       // the chained step's event has not been waited for yet: wait, and bring the host-side mirror of the scalars in step
    auto & A = c->ahead;
    if (!A.ev_pending) return;
    A.ev_pending = false;
    cllm_set_device(c->device);
    cllm_event_sync(A.ev);
    std::lock_guard<std::mutex> lock(g_ring.m);
    for (const auto & ss : A.pred) { const int32_t v = ss.ptr == A.ids_ptr ? A.tok_host[0] : ss.val; auto it = g_i32_vals.find(ss.ptr); if (it != g_i32_vals.end()) it->second = v; }
}
void ahead_launch(hip_backend_ctx * c, bool chained) {
    auto & A = c->ahead;
    A.armed = false; A.snap_ready = false; A.chained = false; A.ev_pending = false;
    cllm_set_device(c->device);
    if (!A.ev && cllm_event_create(&A.ev) != CLLM_OK) return;
    if (!A.tok_host) { void * p = nullptr; if (cllm_host_malloc(&p, 64) != CLLM_OK) return; A.tok_host = (int32_t *) p; }
    if (!A.table_dev && cllm_malloc(&A.table_dev, 1024 * 16) != CLLM_OK) return;
    if (!A.tab_host) { void * p = nullptr; if (cllm_host_malloc(&p, 1024 * 16) != CLLM_OK) return; A.tab_host = (decltype(A.tab_host)) p; }
    if (!A.scratch) { if (cllm_malloc(&A.scratch, 4096) != CLLM_OK) return; if (cllm_memset(A.scratch, 0, 4096, nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) return; }
    if (!A.rng_dev && cllm_malloc(&A.rng_dev, 16 * 24) != CLLM_OK) return;
    if (!A.rng_host) { void * p = nullptr; if (cllm_host_malloc(&p, 16 * 24) != CLLM_OK) return; A.rng_host = (decltype(A.rng_host)) p; }
    size_t snap_need = 0;
    for (auto & o : A.outs) { o.snap_off = snap_need; snap_need += (o.bytes + 255) & ~(size_t) 255; }
    if (A.snap_bytes < snap_need) {
        if (A.snap) { cllm_stream_sync(c->stream); cllm_free(A.snap); A.snap = nullptr; A.snap_bytes = 0; A.rng_cur.clear(); }
        if (cllm_malloc(&A.snap, snap_need) != CLLM_OK) return;
        A.snap_bytes = snap_need;
    }
    // the scalars of the next step as ABSOLUTE values (the position scalars + 1; the token id comes from the argmax): the device words are not read -- ggml-alloc may
    // have handed their blocks to later nodes of the graph that just ran.  One record per distinct pointer (the host's last write of a step wins).
    A.pred.clear();
    int n_rec = 0;
    for (const auto & ss : A.last_sets) {
        A.pred.push_back(ss);
        if (ss.ptr == A.ids_ptr) continue;
        A.pred.back().val = ss.val + 1;
        int k = 0; while (k < n_rec && A.tab_host[k].ptr != ss.ptr) k++;
        if (k == n_rec) { if (n_rec >= 1024) return; n_rec++; }
        A.tab_host[k].ptr = ss.ptr; A.tab_host[k].val = ss.val + 1; A.tab_host[k].pad = 0;
    }
    void * st = c->stream;
    // CLLM_HIP_AHEAD_TIMING=1 (diagnosis): per step, on the GPU's clock -- the captured step alone, what stands in front of it (table copy, prep launch, event), and the token
    // period; a ring of four event triples, read two launches later (the host has synchronized on a later event by then, also with the chain on)
    static void * t_p[4] = {}, * t_g0[4] = {}, * t_g1[4] = {}; static long t_n = 0; static double acc_graph = 0, acc_prep = 0, acc_period = 0; static long acc_n = 0;
    static const bool timing = getenv("CLLM_HIP_AHEAD_TIMING") != nullptr;
    const int t_k = (int)(t_n & 3);
    if (timing) {
        if (!t_p[0]) for (int i = 0; i < 4; i++) { cllm_event_create(&t_p[i]); cllm_event_create(&t_g0[i]); cllm_event_create(&t_g1[i]); }
        if (t_n >= 3) {
            const int a = (int)((t_n - 3) & 3), b = (int)((t_n - 2) & 3);
            float g = 0, q = 0, per = 0;
            if (cllm_event_elapsed_ms(t_g0[a], t_g1[a], &g) == CLLM_OK && cllm_event_elapsed_ms(t_p[a], t_g0[a], &q) == CLLM_OK && cllm_event_elapsed_ms(t_g0[a], t_g0[b], &per) == CLLM_OK) { acc_graph += g; acc_prep += q; acc_period += per; acc_n++; }
            if (acc_n == 64) { HIPB_LOG("ahead timing over 64 steps (GPU clock): token period %.1f us = captured step %.1f + table copy / prep launch / event in front of it %.1f + idle %.1f", acc_period / 64 * 1e3, acc_graph / 64 * 1e3, acc_prep / 64 * 1e3, (acc_period - acc_graph - acc_prep) / 64 * 1e3); acc_graph = acc_prep = acc_period = 0; acc_n = 0; }
        }
        cllm_event_record(t_p[t_k], st);
    }
    if (n_rec && cllm_memcpy_h2d(A.table_dev, A.tab_host, (size_t) n_rec * 16, st) != CLLM_OK) return;      // queued on the step's stream from page-locked memory; done before the event below
    // the snapshot of the outputs, the arg-max and the scalar updates: ONE launch (cllm_op_snapshot_argmax_set; CLLM_HIP_AHEAD_ONE=0: round 5's copies + two launches)
    static const bool one_launch = !getenv("CLLM_HIP_AHEAD_ONE") || atoi(getenv("CLLM_HIP_AHEAD_ONE")) != 0;
    bool fused_prep = one_launch && A.outs.size() <= 16;
    for (const auto & o : A.outs) fused_prep = fused_prep && o.bytes % 4 == 0;
    if (fused_prep) {
        bool same = A.rng_cur.size() == A.outs.size();
        for (size_t k = 0; same && k < A.outs.size(); k++) same = A.rng_cur[k].src == (const void *) A.outs[k].ptr && A.rng_cur[k].dst == (void *)((char *) A.snap + A.outs[k].snap_off) && A.rng_cur[k].bytes == A.outs[k].bytes;
        if (!same) {                               // (the ranges move only when ggml-alloc moves the outputs: the table is uploaded once per captured graph)
            cllm_stream_sync(st);                  // a queued launch may still read the table
            A.rng_cur.clear();
            for (size_t k = 0; k < A.outs.size(); k++) { A.rng_cur.push_back({ (const void *) A.outs[k].ptr, (void *)((char *) A.snap + A.outs[k].snap_off), (uint64_t) A.outs[k].bytes }); A.rng_host[k] = A.rng_cur.back(); }
            if (cllm_memcpy_h2d(A.rng_dev, A.rng_host, A.outs.size() * 24, st) != CLLM_OK) { A.rng_cur.clear(); return; }
        }
    } else for (const auto & o : A.outs) if (cllm_memcpy_d2d((char *) A.snap + o.snap_off, o.ptr, o.bytes, st) != CLLM_OK) { cllm_stream_sync(st); return; }
    // From here on the arg-max below may already have written the token id -- into a block ggml-alloc may have made part of the logits.  If anything fails now the
    // step does not run ahead, but the host's pending read must still see the logits as they were: the snapshot (complete once the stream is idle) serves it.
    auto fail_behind_the_snapshot = [&]() { if (cllm_stream_sync(st) == CLLM_OK) A.snap_ready = true; };
    if (fused_prep) {
        if (cllm_op_snapshot_argmax_set(st, A.rng_dev, (int) A.outs.size(), (const float *) A.logits_ptr, (int64_t)(A.logits_bytes / 4), (int32_t *) A.ids_ptr, A.tok_host, A.table_dev, n_rec, A.scratch) != CLLM_OK) { cllm_stream_sync(st); return; }
    } else
    if (cllm_op_argmax_set(st, (const float *) A.logits_ptr, (int64_t)(A.logits_bytes / 4), (int32_t *) A.ids_ptr, A.tok_host, A.table_dev, n_rec, A.scratch) != CLLM_OK) { fail_behind_the_snapshot(); return; }
    if (cllm_event_record(A.ev, st) != CLLM_OK) { fail_behind_the_snapshot(); return; }
    if (timing) cllm_event_record(t_g0[t_k], st);
    if ((A.tp ? tp_ahead_replay(c) : cllm_graph_launch(c->graph_exec, st)) != CLLM_OK) { fail_behind_the_snapshot(); return; }
    if (timing) { cllm_event_record(t_g1[t_k], st); t_n++; }
    static const bool ahead_sync = getenv("CLLM_HIP_AHEAD_SYNC") != nullptr;       // (debugging: run the step ahead to completion before returning)
    if (ahead_sync) cllm_stream_sync(st);
    A.inflight = true; A.launched++; A.chained = chained; A.ev_pending = true;
    // not chained: the snapshot and the scalars are in place before the host goes on (its own writes of the same scalars come later); the device then holds the predicted
    // scalars: the host-side mirror is brought in step (the host will write the same values again).  Chained: synchronize() does both.
    if (!chained) ahead_finish_event(c);
}


// a step is running ahead: is it the one the host is asking for?  1: yes (nothing to launch), 0: no -- the step was allowed to finish and the host's scalars are in place (buf_set
// held them back), -1: error.  sig_equal: the host's launch list equals the captured one the step was started with.
int ahead_resolve(hip_backend_ctx * c, bool sig_equal, const std::vector<hip_backend_ctx::scalar_set> & cur_sets) {
    auto & A = c->ahead;
    ahead_finish_event(c);                 // (a host that did not synchronize since: the predicted token must be known before the comparison below)
    A.inflight = false;
    bool ok = sig_equal && cur_sets.size() == A.pred.size();
    for (size_t k = 0; ok && k < cur_sets.size(); k++)
        ok = cur_sets[k].ptr == A.pred[k].ptr && cur_sets[k].val == (A.pred[k].ptr == A.ids_ptr ? A.tok_host[0] : A.pred[k].val);
    if (ok) { A.hits++; A.misses = 0; return 1; }
    static const bool adbg = getenv("CLLM_HIP_AHEAD_DEBUG") != nullptr;
    if (adbg) {
        HIPB_LOG("ahead: MISS sig_equal=%d sets %zu predicted %zu", (int) sig_equal, cur_sets.size(), A.pred.size());
        for (size_t k = 0; k < cur_sets.size() && k < A.pred.size(); k++) {
            const int32_t want = A.pred[k].ptr == A.ids_ptr ? A.tok_host[0] : A.pred[k].val;
            if (cur_sets[k].ptr != A.pred[k].ptr || cur_sets[k].val != want)
                HIPB_LOG("ahead:   set %zu: host wrote %p = %d, predicted %p = %d%s", k, cur_sets[k].ptr, (int) cur_sets[k].val, A.pred[k].ptr, (int) want, A.pred[k].ptr == A.ids_ptr ? " (token id)" : "");
        }
    }
    if (++A.misses >= 2) { A.skip_until = A.graphs + 64; A.misses = 0; }
    cllm_set_device(c->device);
    cllm_stream_sync(c->stream);           // the step that ran ahead is not the one asked for: let it finish (every rank's stream joins this one), then put the host's scalars in place
    for (const auto & ss : cur_sets) if (cllm_memcpy_h2d((void *) ss.ptr, &ss.val, 4, nullptr) != CLLM_OK) { HIPB_LOG("scalar write failed: %s", cllm_last_error()); return -1; }
    if (cllm_stream_sync(nullptr) != CLLM_OK) return -1;
    return 0;
}

// ---- tensor parallel BEHIND the boundary (CLLM_HIP_TP=N): ONE logical ggml device over N ranks ---------------------------------------------------------------
// The reference's own slot for this is SplitMethod::Row -- "TODO: WIP" (src/backend.h:322-327; device assignment src/backend.cpp:677-778 only splits by layer, which at
// batch 1 buys capacity, not speed).  Here the unmodified host sees a single device "HIP0"; its buffers -- weights, KV caches, activations -- live in rank 0's HBM exactly as
// on one GPU, and every graph that is not a recognised decode step (prompts, anything unusual) runs there un-sharded, bit-identical to the single-device module.  A graph that IS
// the five-launch-per-layer decode step (GET_ROWS, then per layer q|k|v -> attention -> o + residual -> gate/up -> down + residual, then norm + lm_head: the same patterns
// make_plan fuses) runs tensor-parallel:
//   * shards are cut ON THE DEVICE from the tensors the host uploaded, the first time a step needs them (like the packed copies above; validity = the source buffers' generation
//     counters): q / k / v and gate / up by ROWS (whole KV groups; F by the down projection's quant blocks, cllm_tp_split, uneven allowed), o / down by whole quant blocks of K;
//   * every rank keeps the residual stream itself; o / down send their partial rows as granules into every rank's receive buffer and the next RMS_NORM mat-vec adds them in rank
//     order (cllm_op_mul_mat_vec_tp_scatter / _gather: NO all-reduce launch, five launches per layer and rank as on one GPU); the lm_head runs on rank 0 into the host's logits;
//   * one KV shard per rank (dense, the rank's KV heads only), refreshed from the host's cache after any un-sharded graph or buffer write, and written back row by row after
//     every step (cllm_op_kv_shard_copy), so the host's cache stays what the reference expects.
// Ranks on distinct GPUs run on their own streams (the gathers poll for the peers' granules); ranks that share a GPU (N > GPUs: the one-GPU test vehicle) share rank 0's stream and
// are issued site by site, so a gather never waits for a launch that cannot start.  Everything joins rank 0's stream at the end of the step: synchronize() is unchanged.
// Results: the fp32 sums of o / down are split into N partial chains -> tolerance tier (SURVEY 8e "T1/T2"), NOT the bit-exact tier of the single device.
struct tp_layer { int gq = -1, attn = -1, o = -1, ggu = -1, down = -1; };
struct tp_desc { int embed = -1, head = -1, head_norm = -1, head_mm = -1; std::vector<tp_layer> layers; };      // head: a fused norm + mat-vec (mvs entry), or the two nodes

// is this graph the decode step the tensor-parallel path takes?  (node order: GET_ROWS, L x { q|k|v group, attention level 2, o + residual, gate/up group, down + residual }, head)
bool tp_extract(ggml_cgraph * g, const fuse_plan & P, tp_desc & D) {
    enum { S_EMBED, S_QKV, S_ATTN, S_O, S_GU, S_DOWN, S_HEAD_MM, S_DONE } st = S_EMBED;
    std::vector<uint8_t> seen(P.groups.size(), 0);
    tp_layer cur;
    for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        if (ggml_is_empty(t) || P.skip[i]) continue;
        if (t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) continue;
        if (t->op == GGML_OP_GET_ROWS) { if (st != S_EMBED) return false; D.embed = i; st = S_QKV; continue; }
        // the head as chatllm builds it (LMFinalSteps, src/models.cpp:1736-1784): the normalised hidden state is a graph OUTPUT, so the norm stays a launch of its own
        if (t->op == GGML_OP_MUL && P.alt[i] == ALT_RMS_NORM_MUL && st == S_QKV && !D.layers.empty()) { D.head_norm = i; st = S_HEAD_MM; continue; }
        if (t->op == GGML_OP_MUL_MAT && st == S_HEAD_MM && P.mv[i] < 0 && P.pf[i] < 0 && P.alt[i] == ALT_NONE) {
            const ggml_tensor * hn = ggml_graph_node(g, D.head_norm);
            if (!t->src[1] || t->src[1]->data != hn->data || ggml_nelements(t->src[1]) != hn->ne[0] || !f32_vec(t) || !is_q(t->src[0]->type)) return false;
            D.head_mm = i; st = S_DONE; continue;
        }
        if (t->op == GGML_OP_MUL_MAT && P.mv[i] >= 0 && P.alt[i] == ALT_NONE) {
            const fused_mv & f = P.mvs[P.mv[i]];
            if (f.group >= 0) {
                if (seen[f.group]) continue;                  // a later member of a group already taken
                seen[f.group] = 1;
                const merge_group & G = P.groups[f.group];
                if (!G.interleave && G.n == 3 && st == S_QKV) { cur = tp_layer(); cur.gq = f.group; st = S_ATTN; continue; }
                if (G.interleave && G.n == 2 && st == S_GU) { cur.ggu = f.group; st = S_DOWN; continue; }
                return false;
            }
            if (f.pro == 2 && f.resid && st == S_O) { cur.o = P.mv[i]; st = S_GU; continue; }
            if (f.pro == 4 && f.resid && st == S_DOWN && P.groups[cur.ggu].consumer == P.mv[i]) { cur.down = P.mv[i]; D.layers.push_back(cur); st = S_QKV; continue; }
            if (f.pro == 1 && !f.resid && st == S_QKV && !D.layers.empty()) { D.head = P.mv[i]; st = S_DONE; continue; }
            return false;
        }
        if ((t->op == GGML_OP_CPY || t->op == GGML_OP_DUP || t->op == GGML_OP_CONT) && P.attn[i] >= 0 && st == S_ATTN && P.attns[P.attn[i]].level == 2) { cur.attn = P.attn[i]; st = S_O; continue; }
        return false;
    }
    return st == S_DONE;
}

struct tp_dims { int64_t H = 0, F = 0, hd = 0; int nh = 0, nkv = 0, gs = 0; int64_t ML = 0; };
struct tp_split { int64_t kv0 = 0, kvc = 0, f0 = 0, fc = 0; };        // this rank's KV heads [kv0, kv0 + kvc) and FFN features [f0, f0 + fc)

#define FAIL_TP(msg) do { HIPB_LOG("tensor parallel: %s", msg); return CLLM_E_UNSUPPORTED; } while (0)
int tp_ensure_group(hip_backend_ctx * c, int n_sites, size_t max_n) {
    auto & T = g_tp;
    if (!T.r.empty() && T.owner != c) FAIL_TP("another backend context of this process owns the tensor-parallel ranks");
    if (T.r.empty()) {
        T.owner = c;
        const int phys = cllm_device_count();
        T.r.resize(T.n);
        for (int k = 0; k < T.n; k++) {
            tp_rank & R = T.r[k];
            R.gpu = (c->device + k) % (phys > 0 ? phys : 1);
            // CLLM_HIP_TP_STREAMS=1 (tests): ranks that share rank 0's GPU get streams of their own as well -- the event / cross-stream path of distinct GPUs on a one-GPU box.
            // Only for shapes whose launches can all be resident at once (a gather polls for scatters of other streams): the test shapes; bounded waits otherwise report a time-out.
            static const bool force_streams = getenv("CLLM_HIP_TP_STREAMS") && atoi(getenv("CLLM_HIP_TP_STREAMS")) != 0;
            if (R.gpu == c->device && !(force_streams && k > 0)) R.stream = c->stream;
            else {
                cllm_set_device(R.gpu);
                if (int rc = cllm_stream_create(&R.stream)) { cllm_set_device(c->device); return rc; }
                if (int rc = cllm_event_create(&R.ev)) { cllm_set_device(c->device); return rc; }
                R.own_stream = true;
            }
        }
        cllm_set_device(c->device);
        if (int rc = cllm_event_create(&T.ev0)) return rc;
    }
    if (T.fused_ready && (T.sites < n_sites || T.max_n < max_n)) {              // another model on the same logical device: bigger buffers
        for (tp_rank & R : T.r) { cllm_set_device(R.gpu); cllm_stream_sync(R.stream); if (R.fused) cllm_tp_fused_destroy(R.fused); R.fused = nullptr; }
        cllm_set_device(c->device);
        T.fused_ready = false;
    }
    if (!T.fused_ready) {
        std::vector<int> devs(T.n); std::vector<void *> os(T.n, nullptr);
        for (int k = 0; k < T.n; k++) devs[k] = T.r[k].gpu;
        const size_t mn = (max_n + 3) & ~(size_t) 3;
        if (int rc = cllm_tp_fused_create_group(T.n, devs.data(), n_sites, mn, os.data())) return rc;
        for (int k = 0; k < T.n; k++) T.r[k].fused = os[k];
        T.sites = n_sites; T.max_n = mn; T.fused_ready = true;
    }
    return CLLM_OK;
}

// rank k's shard made of 2-D pieces of up to three source tensors; built on rank 0's GPU (where the sources live), moved to the rank's GPU if that is another one
void * tp_get_shard(hip_backend_ctx * c, int k, const ggml_tensor * const * srcs, int n, const tp_piece * pieces, int np, size_t bytes, bool * built) {
    tp_rank & R = g_tp.r[k];
    for (int i = 0; i < n; i++) if (!ours(srcs[i]) || ((const hip_buffer_ctx *) srcs[i]->buffer->context)->device != c->device) return nullptr;
    tp_shard & e = R.shards[srcs[0]->data];
    bool valid = e.data && e.n == n && e.bytes == bytes;
    for (int i = 0; i < n && valid; i++) { const auto * bc = (const hip_buffer_ctx *) srcs[i]->buffer->context; valid = e.src[i] == srcs[i]->data && e.uid[i] == bc->uid && e.gen[i] == bc->gen.load(); }
    if (valid) return e.data;
    if (e.data) { cllm_set_device(R.gpu); cllm_stream_sync(R.stream); cllm_free(e.data); cllm_set_device(c->device); }
    e = tp_shard(); e.n = n; e.bytes = bytes;
    for (int i = 0; i < n; i++) { const auto * bc = (const hip_buffer_ctx *) srcs[i]->buffer->context; e.src[i] = srcs[i]->data; e.uid[i] = bc->uid; e.gen[i] = bc->gen.load(); }
    void * local = nullptr;
    cllm_set_device(c->device);
    if (cllm_malloc(&local, bytes ? bytes : 1) != CLLM_OK) return nullptr;
    for (int i = 0; i < np; i++)
        if (cllm_copy_2d(c->stream, (char *) local + pieces[i].doff, pieces[i].dpitch, pieces[i].src, pieces[i].spitch, pieces[i].width, pieces[i].rows) != CLLM_OK) { cllm_stream_sync(c->stream); cllm_free(local); return nullptr; }
    if (R.gpu == c->device) e.data = local;
    else {
        void * remote = nullptr;
        cllm_set_device(R.gpu);
        const int rc = cllm_malloc(&remote, bytes ? bytes : 1);
        cllm_set_device(c->device);
        if (rc != CLLM_OK || cllm_memcpy_peer_async(remote, R.gpu, local, c->device, bytes, c->stream) != CLLM_OK || cllm_stream_sync(c->stream) != CLLM_OK) {
            cllm_stream_sync(c->stream); cllm_free(local);
            if (remote) { cllm_set_device(R.gpu); cllm_free(remote); cllm_set_device(c->device); }
            return nullptr;
        }
        cllm_free(local);
        e.data = remote;
    }
    *built = true;
    return e.data;
}

// after a synchronize: did a bounded granule wait of the last tensor-parallel step time out?
bool tp_check_errors() {
    auto & T = g_tp;
    if (T.n <= 1 || !T.last_tp || !T.fused_ready) return true;
    T.last_tp = false;
    bool ok = true;
    for (tp_rank & R : T.r) { cllm_set_device(R.gpu); if (R.fused && cllm_tp_fused_error(R.fused)) ok = false; }
    cllm_set_device(T.r[0].gpu);
    return ok;
}
void tp_free_all() {
    auto & T = g_tp;
    for (tp_rank & R : T.r) {
        cllm_set_device(R.gpu);
        if (R.stream) cllm_stream_sync(R.stream);
        for (auto & kv : R.shards) if (kv.second.data) cllm_free(kv.second.data);
        R.shards.clear();
        for (void * p : R.kv_mem) if (p) cllm_free(p);
        R.kv_mem.clear(); R.kv_sig.clear();
        if (R.kv_table) cllm_free(R.kv_table);
        if (R.scratch) cllm_free(R.scratch);
        if (R.fused) cllm_tp_fused_destroy(R.fused);
        if (R.ev) cllm_event_destroy(R.ev);
        if (R.own_stream && R.stream) cllm_stream_destroy(R.stream);
    }
    for (void * e : T.execs) if (e) cllm_graph_destroy(e);
    T.execs.clear(); T.graph_sig.clear(); T.last_sig.clear();
    if (T.ev0) cllm_event_destroy(T.ev0);
    T.r.clear(); T.fused_ready = false; T.ev0 = nullptr; T.owner = nullptr;
}

// one decode step, tensor-parallel.  GGML_STATUS_ABORTED = "not taken, nothing launched" (the caller runs the graph un-sharded on rank 0)
int ahead_resolve(hip_backend_ctx * c, bool sig_equal, const std::vector<hip_backend_ctx::scalar_set> & cur_sets);
ggml_status tp_graph_compute(hip_backend_ctx * c, ggml_cgraph * g, fuse_plan & P, const std::vector<hip_backend_ctx::scalar_set> & cur_sets, bool * replayed_out) {
    auto & T = g_tp;
    tp_desc D;
    static const bool tp_dbg = getenv("CLLM_HIP_TP_DEBUG") != nullptr;
#define TP_NO() (tp_dbg ? (fprintf(stderr, "[ggml-hip] tensor parallel: decode step not taken (ggml-hip.cpp:%d): un-sharded on rank 0\n", __LINE__), GGML_STATUS_ABORTED) : GGML_STATUS_ABORTED)
    if (T.broken) return GGML_STATUS_ABORTED;
    if (!tp_extract(g, P, D)) {
        static const bool dbg = getenv("CLLM_HIP_TP_DEBUG") != nullptr;
        if (dbg) HIPB_LOG("tensor parallel: a %d-node graph is not the decode step (%zu layers matched, embed %d, head %d/%d/%d): un-sharded on rank 0", ggml_graph_n_nodes(g), D.layers.size(), D.embed, D.head, D.head_norm, D.head_mm);
        return GGML_STATUS_ABORTED;
    }
    auto node = [&](int i) { return ggml_graph_node(g, i); };
    const int N = T.n, L = (int) D.layers.size();
    // ---- the chain of residual streams and the shapes: everything is checked BEFORE the first launch ----
    const ggml_tensor * emb = node(D.embed);
    if (!f32_vec(emb)) return TP_NO();
    tp_dims d; d.H = emb->ne[0];
    const float * curx = (const float *) emb->data;
    const fused_attn & A0 = P.attns[D.layers[0].attn];
    d.hd = A0.hd; d.nh = A0.nh; d.nkv = A0.nkv; d.ML = A0.ML;
    if (d.nkv < N || d.nh % d.nkv) return TP_NO();
    d.gs = d.nh / d.nkv;
    struct lw { const ggml_tensor * wq, * wk, * wv, * bq, * bk, * bv, * wo, * wg, * wu, * wd; const float * an, * fn; float an_eps, fn_eps; };
    std::vector<lw> W((size_t) L);
    for (int l = 0; l < L; l++) {
        const tp_layer & Y = D.layers[l];
        const merge_group & G = P.groups[Y.gq]; const fused_attn & A = P.attns[Y.attn];
        if (G.member[0] != A.wq || G.member[1] != A.wk || G.member[2] != A.wv) return TP_NO();
        const fused_mv & fq = P.mvs[A.wq], & fk = P.mvs[A.wk], & fv = P.mvs[A.wv], & fo = P.mvs[Y.o], & fd = P.mvs[Y.down];
        const merge_group & GG = P.groups[Y.ggu];
        const fused_mv & fg = P.mvs[GG.member[0]], & fu = P.mvs[GG.member[1]];
        lw & w = W[(size_t) l];
        w.wq = node(fq.node)->src[0]; w.wk = node(fk.node)->src[0]; w.wv = node(fv.node)->src[0]; w.wo = node(fo.node)->src[0];
        w.wg = node(fg.node)->src[0]; w.wu = node(fu.node)->src[0]; w.wd = node(fd.node)->src[0];
        w.bq = G.bias ? fq.resid_t : nullptr; w.bk = G.bias ? fk.resid_t : nullptr; w.bv = G.bias ? fv.resid_t : nullptr;
        w.an = fq.pw; w.an_eps = fq.eps; w.fn = fg.pw; w.fn_eps = fg.eps;
        if (A.hd != d.hd || A.nh != d.nh || A.nkv != d.nkv || A.ML != d.ML || A.mode != A0.mode || A.freq_base != A0.freq_base || A.n_kv != A0.n_kv) return TP_NO();
        if (fq.px != curx || fo.px != A.out || fo.resid != curx) return TP_NO();
        curx = fo.dst;
        if (fg.px != curx || fd.resid != curx) return TP_NO();
        curx = fd.dst;
        if (w.wq->ne[0] != d.H || w.wq->ne[1] != d.nh * d.hd || w.wk->ne[1] != d.nkv * d.hd || w.wv->ne[1] != d.nkv * d.hd || w.wo->ne[0] != d.nh * d.hd || w.wo->ne[1] != d.H) return TP_NO();
        if (l == 0) d.F = w.wg->ne[1];
        if (w.wg->ne[0] != d.H || w.wg->ne[1] != d.F || w.wu->ne[1] != d.F || w.wd->ne[0] != d.F || w.wd->ne[1] != d.H || w.wg->type != w.wu->type) return TP_NO();
        if (w.wq->type != W[0].wq->type || w.wo->type != W[0].wo->type || w.wg->type != W[0].wg->type || w.wd->type != W[0].wd->type) return TP_NO();
    }
    const ggml_tensor * wh = nullptr;
    if (D.head >= 0) {
        const fused_mv & fh = P.mvs[D.head];
        if (fh.px != curx) return TP_NO();
        wh = node(fh.node)->src[0];
    } else {
        const ggml_tensor * hn = node(D.head_norm);
        if (!hn->src[0]->src[0] || hn->src[0]->src[0]->data != (const void *) curx || ggml_nelements(hn) != d.H) return TP_NO();
        wh = node(D.head_mm)->src[0];
    }
    auto kind = [](ggml_type t) { return (int64_t)(t == GGML_TYPE_Q4_K ? 256 : 32); };
    const ggml_type tq = W[0].wq->type, to = W[0].wo->type, tg = W[0].wg->type, td = W[0].wd->type;
    const bool mv_head_ok = is_q(wh->type) && wh->ne[2] == 1 && wh->ne[3] == 1 && wh->nb[1] == ggml_row_size(wh->type, wh->ne[0]) && !((uintptr_t) wh->data & 15) && !(wh->nb[1] & 15) && wh->ne[0] == d.H;
    // what the two tensor-parallel forms take (gemv_tp.hip) -- and the whole-KV-group / whole-quant-block splits
    if (d.H % kind(tq) || d.H % kind(tg) || d.H % kind(wh->type) || d.H > 16384 || d.H % 4) return TP_NO();
    const int64_t fblk = ggml_blck_size(td), nfb = d.F / fblk;
    if (d.F % fblk || nfb < N || d.F % 8) return TP_NO();
    std::vector<tp_split> S((size_t) N);
    for (int k = 0; k < N; k++) {
        tp_split & s = S[(size_t) k];
        int64_t b0, bc;
        cllm_tp_split(d.nkv, N, k, &s.kv0, &s.kvc); cllm_tp_split(nfb, N, k, &b0, &bc);
        s.f0 = b0 * fblk; s.fc = bc * fblk;
        const int64_t qc = s.kvc * d.gs * d.hd, q0 = s.kv0 * d.gs * d.hd;
        if (s.kvc < 1 || qc % kind(to) || q0 % ggml_blck_size(to) || qc > 32768 || s.fc % kind(td) || s.fc > 32768 || (s.fc % 8) ||
            !cllm_attn_decode_supported((int)(s.kvc * d.gs), (int) s.kvc, (int) d.hd, d.ML)) return TP_NO();
    }
    if (tp_ensure_group(c, 2 * L, (size_t) d.H) != CLLM_OK) { HIPB_LOG("tensor parallel: %s -- running un-sharded on rank 0 from now on", cllm_last_error()); T.broken = true; cllm_set_device(c->device); return TP_NO(); }
    if (!T.told) { T.told = true; HIPB_LOG("tensor parallel: %d ranks behind one ggml device (%d layers; KV heads / FFN features of rank 0: %lld / %lld of %d / %lld)", N, L, (long long) S[0].kvc, (long long) S[0].fc, d.nkv, (long long) d.F); }

    // ---- shards (first step, or after the host rewrote a weight) ----
    struct ls { void * qkv, * bias, * o, * gu, * dn; const float * an, * fn; };
    std::vector<std::vector<ls>> SH((size_t) N, std::vector<ls>((size_t) L));
    bool built = false;
    auto fail_shard = [&]() { HIPB_LOG("tensor parallel: a weight shard could not be made (%s) -- running un-sharded on rank 0 from now on", cllm_last_error()); T.broken = true; cllm_set_device(c->device); return TP_NO(); };
    auto replicate = [&](int k, const float * p, size_t bytes, const ggml_tensor * owner) -> const float * {      // a small F32 tensor every rank reads (norm weights)
        if (T.r[(size_t) k].gpu == c->device) return p;
        const ggml_tensor * src[1] = { owner };
        tp_piece pc = { p, bytes, bytes, 1, 0, bytes };
        return (const float *) tp_get_shard(c, k, src, 1, &pc, 1, bytes, &built);
    };
    auto norm_owner = [&](const fused_mv & f) { return node(f.node)->src[1]->src[1]; };                        // MUL_MAT(w, MUL(RMS_NORM(x), weight)): the weight tensor
    for (int k = 0; k < N; k++) {
        const tp_split & s = S[(size_t) k];
        const int64_t qr0 = s.kv0 * d.gs * d.hd, qrc = s.kvc * d.gs * d.hd, kr0 = s.kv0 * d.hd, krc = s.kvc * d.hd;
        for (int l = 0; l < L; l++) {
            const lw & w = W[(size_t) l]; ls & o = SH[(size_t) k][(size_t) l];
            const tp_layer & Y = D.layers[l];
            {   // q | k | v rows of this rank's KV groups
                const size_t rb = w.wq->nb[1];
                const ggml_tensor * src[3] = { w.wq, w.wk, w.wv };
                tp_piece pc[3] = { { (const char *) w.wq->data + qr0 * rb, qrc * rb, qrc * rb, 1, 0, qrc * rb }, { (const char *) w.wk->data + kr0 * rb, krc * rb, krc * rb, 1, (size_t) qrc * rb, krc * rb },
                                   { (const char *) w.wv->data + kr0 * rb, krc * rb, krc * rb, 1, (size_t)(qrc + krc) * rb, krc * rb } };
                if (w.wk->nb[1] != rb || w.wv->nb[1] != rb || !(o.qkv = tp_get_shard(c, k, src, 3, pc, 3, (size_t)(qrc + 2 * krc) * rb, &built))) return fail_shard();
                o.bias = nullptr;
                if (w.bq) {
                    const ggml_tensor * bs[3] = { w.bq, w.bk, w.bv };
                    tp_piece bp[3] = { { (const char *) w.bq->data + qr0 * 4, (size_t) qrc * 4, (size_t) qrc * 4, 1, 0, (size_t) qrc * 4 }, { (const char *) w.bk->data + kr0 * 4, (size_t) krc * 4, (size_t) krc * 4, 1, (size_t) qrc * 4, (size_t) krc * 4 },
                                       { (const char *) w.bv->data + kr0 * 4, (size_t) krc * 4, (size_t) krc * 4, 1, (size_t)(qrc + krc) * 4, (size_t) krc * 4 } };
                    if (!(o.bias = tp_get_shard(c, k, bs, 3, bp, 3, (size_t)(qrc + 2 * krc) * 4, &built))) return fail_shard();
                }
            }
            {   // o: the quant blocks of K that belong to this rank's heads
                const size_t wb = ggml_row_size(to, qrc), ob = ggml_row_size(to, qr0);
                const ggml_tensor * src[1] = { w.wo };
                tp_piece pc = { (const char *) w.wo->data + ob, w.wo->nb[1], wb, (size_t) d.H, 0, wb };
                if (!(o.o = tp_get_shard(c, k, src, 1, &pc, 1, wb * (size_t) d.H, &built))) return fail_shard();
            }
            {   // gate / up rows [f0, f0 + fc), alternating
                const size_t rb = w.wg->nb[1];
                const ggml_tensor * src[2] = { w.wg, w.wu };
                tp_piece pc[2] = { { (const char *) w.wg->data + s.f0 * rb, rb, rb, (size_t) s.fc, 0, 2 * rb }, { (const char *) w.wu->data + s.f0 * rb, rb, rb, (size_t) s.fc, rb, 2 * rb } };
                if (w.wu->nb[1] != rb || !(o.gu = tp_get_shard(c, k, src, 2, pc, 2, 2 * (size_t) s.fc * rb, &built))) return fail_shard();
            }
            {   // down: the quant blocks [f0 / blk, (f0 + fc) / blk) of every row
                const size_t wb = ggml_row_size(td, s.fc), ob = ggml_row_size(td, s.f0);
                const ggml_tensor * src[1] = { w.wd };
                tp_piece pc = { (const char *) w.wd->data + ob, w.wd->nb[1], wb, (size_t) d.H, 0, wb };
                if (!(o.dn = tp_get_shard(c, k, src, 1, &pc, 1, wb * (size_t) d.H, &built))) return fail_shard();
            }
            o.an = replicate(k, w.an, (size_t) d.H * 4, norm_owner(P.mvs[P.attns[Y.attn].wq]));
            o.fn = replicate(k, w.fn, (size_t) d.H * 4, norm_owner(P.mvs[P.groups[Y.ggu].member[0]]));
            if (!o.an || !o.fn) return fail_shard();
        }
    }
    // ---- per-rank scratch: [cos/sin 1 KB][pos 256 B][x ping][x pong][q|k|v][attention out][SiLU*up][scores of the long-context attention] ----
    auto al = [](size_t b) { return (b + 255) & ~(size_t) 255; };
    const bool table = d.hd == 64 || d.hd == 128;
    for (int k = 0; k < N; k++) {
        tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k];
        const size_t need = 1024 + 256 + 2 * al((size_t) d.H * 4) + al((size_t)(s.kvc * (d.gs + 2) * d.hd) * 4) + al((size_t)(s.kvc * d.gs * d.hd) * 4) + al((size_t) s.fc * 4) +
                            al(cllm_attn_decode_wsize(A0.n_kv, (int)(s.kvc * d.gs), d.ML));
        if (need > R.scratch_bytes) {
            cllm_set_device(R.gpu); cllm_stream_sync(R.stream);
            if (R.scratch) cllm_free(R.scratch);
            R.scratch = nullptr; R.scratch_bytes = 0;
            void * p = nullptr;
            if (cllm_malloc(&p, need + need / 4) != CLLM_OK) { cllm_set_device(c->device); return GGML_STATUS_ALLOC_FAILED; }
            R.scratch = (char *) p; R.scratch_bytes = need + need / 4;
        }
    }
    // ---- KV shards: table of { host K, host V, shard K, shard V } per layer on every rank's GPU ----
    const uint64_t epoch = g_tp_kv_epoch.load();
    for (int k = 0; k < N; k++) {
        tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k];
        std::vector<const void *> sig; sig.reserve((size_t) 2 * L);
        for (int l = 0; l < L; l++) { const fused_attn & A = P.attns[D.layers[l].attn]; sig.push_back(A.k_cache); sig.push_back(A.v_cache); }
        const size_t kd = (size_t)(s.kvc * d.hd);
        if (sig != R.kv_sig || R.kv_dims[0] != kd || R.kv_dims[1] != (size_t) d.ML || R.kv_dims[2] != (size_t) s.kv0) {
            cllm_set_device(R.gpu); cllm_stream_sync(R.stream);
            for (void * p : R.kv_mem) if (p) cllm_free(p);
            R.kv_mem.assign((size_t) 2 * L, nullptr);
            if (!R.kv_table && cllm_malloc(&R.kv_table, 4096 * 32) != CLLM_OK) { cllm_set_device(c->device); return GGML_STATUS_ALLOC_FAILED; }
            if (L > 4096) { cllm_set_device(c->device); return TP_NO(); }
            std::vector<void *> tab((size_t) 4 * L);
            for (int l = 0; l < L; l++) {
                for (int h = 0; h < 2; h++) if (cllm_malloc(&R.kv_mem[(size_t)(2 * l + h)], kd * (size_t) d.ML * 2) != CLLM_OK) { cllm_set_device(c->device); return GGML_STATUS_ALLOC_FAILED; }
                tab[(size_t) 4 * l] = (void *) sig[(size_t) 2 * l]; tab[(size_t) 4 * l + 1] = (void *) sig[(size_t) 2 * l + 1]; tab[(size_t) 4 * l + 2] = R.kv_mem[(size_t) 2 * l]; tab[(size_t) 4 * l + 3] = R.kv_mem[(size_t) 2 * l + 1];
            }
            if (cllm_memcpy_h2d(R.kv_table, tab.data(), tab.size() * sizeof(void *), nullptr) != CLLM_OK || cllm_stream_sync(nullptr) != CLLM_OK) { cllm_set_device(c->device); return GGML_STATUS_FAILED; }
            R.kv_sig = sig; R.kv_dims[0] = kd; R.kv_dims[1] = (size_t) d.ML; R.kv_dims[2] = (size_t) s.kv0; R.kv_valid = 0;
        }
        if (R.kv_epoch != epoch) { R.kv_valid = 0; R.kv_epoch = epoch; }
    }
    cllm_set_device(c->device);
    if (built && cllm_stream_sync(c->stream) != CLLM_OK) return GGML_STATUS_FAILED;

    // ---- the step ----
    void * st0 = c->stream;
#define TPC(expr) do { const int rc_ = (expr); if (rc_ != CLLM_OK) { HIPB_LOG("tensor-parallel step: %s failed: %s", #expr, cllm_last_error()); cllm_set_device(c->device); return rc_ == CLLM_E_ALLOC ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_FAILED; } } while (0)
    struct rk { float * cs, * xa, * xb, * qkv, * att, * act; int32_t * pos; void * scores; size_t score_bytes; float * cur; };
    std::vector<rk> X((size_t) N);
    for (int k = 0; k < N; k++) {                 // where every rank keeps its inputs and intermediates (pointer arithmetic only: nothing is launched before the signature is known)
        tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k];
        char * b = R.scratch;
        x.cs = (float *) b; b += 1024; x.pos = (int32_t *) b; b += 256;
        x.xa = (float *) b; b += al((size_t) d.H * 4); x.xb = (float *) b; b += al((size_t) d.H * 4);
        x.qkv = (float *) b; b += al((size_t)(s.kvc * (d.gs + 2) * d.hd) * 4); x.att = (float *) b; b += al((size_t)(s.kvc * d.gs * d.hd) * 4);
        x.act = (float *) b; b += al((size_t) s.fc * 4);
        x.score_bytes = cllm_attn_decode_wsize(A0.n_kv, (int)(s.kvc * d.gs), d.ML); x.scores = x.score_bytes ? (void *) b : nullptr;
        x.cur = x.xa;
    }
    // what stands in front of the sharded launches: the embedding row on rank 0, that row and the position handed to every rank (queued on rank 0's stream: behind the GET_ROWS and
    // behind everything an earlier un-sharded graph wrote), the ranks' streams made to wait for them, the cache shards brought up to date
    auto pre_step = [&]() -> ggml_status {
        cllm_set_device(c->device);
        cllm_tensor da = desc(emb->src[0]), db = desc(emb->src[1]), dd = desc(emb);
        TPC(cllm_op_get_rows(st0, &da, &db, &dd));
        bool remote = false;
        for (int k = 0; k < N; k++) {
            tp_rank & R = T.r[(size_t) k]; rk & x = X[(size_t) k];
            TPC(cllm_memcpy_peer_async(x.xa, R.gpu, emb->data, c->device, (size_t) d.H * 4, st0));
            TPC(cllm_memcpy_peer_async(x.pos, R.gpu, A0.pos, c->device, 4, st0));
            remote = remote || R.own_stream;
        }
        if (remote) {
            TPC(cllm_event_record(T.ev0, st0));
            for (int k = 0; k < N; k++) if (T.r[(size_t) k].own_stream) { cllm_set_device(T.r[(size_t) k].gpu); TPC(cllm_stream_wait_event(T.r[(size_t) k].stream, T.ev0)); }
        }
        for (int k = 0; k < N; k++) {
            tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k];
            cllm_set_device(R.gpu);
            if (R.kv_valid < A0.n_kv - 1) {          // rows the host's cache has and the shard has not (a prompt ran un-sharded, a session was loaded ...)
                TPC(cllm_op_kv_shard_copy(R.stream, R.kv_table, L, (int)(s.kvc * d.hd), (int)(d.nkv * d.hd), (int)(s.kv0 * d.hd), d.ML, R.kv_valid, A0.n_kv - 1, nullptr, 0));
            }
            R.kv_valid = A0.n_kv;
        }
        return GGML_STATUS_SUCCESS;
    };
    auto wdesc = [](ggml_type t, int64_t K, int64_t rows, void * data) {
        cllm_tensor w; w.type = (int32_t) t; w.ne[0] = K; w.ne[1] = rows; w.ne[2] = w.ne[3] = 1;
        w.nb[0] = ggml_type_size(t); w.nb[1] = ggml_row_size(t, K); w.nb[2] = w.nb[3] = w.nb[1] * (size_t) rows; w.data = data; return w;
    };
    // ---- the head's operands.  Sharded by ROWS of the lm_head (default; CLLM_HIP_TP_HEAD=0: all of it on rank 0): every rank folds the last all-reduce into its own residual
    //      stream and writes its logit rows straight into the host's logits tensor in rank 0's memory (a peer store from another GPU); a row's arithmetic does not change, so the
    //      logits are the bits rank 0 alone would have produced from the same residual stream.  chatllm keeps the normalised hidden state as a graph OUTPUT: rank 0 also runs that node.
    static const bool shard_head = !getenv("CLLM_HIP_TP_HEAD") || atoi(getenv("CLLM_HIP_TP_HEAD")) != 0;
    const fused_mv * fh = D.head >= 0 ? &P.mvs[D.head] : nullptr;
    ggml_tensor * hn = D.head >= 0 ? nullptr : node(D.head_norm), * hm = D.head >= 0 ? nullptr : node(D.head_mm);
    const float * norm_w = fh ? fh->pw : (const float *) hn->src[1]->data;
    const ggml_tensor * norm_t = fh ? norm_owner(*fh) : hn->src[1];
    float eps_h = fh ? fh->eps : 0.0f;
    if (!fh) memcpy(&eps_h, hn->src[0]->op_params, 4);
    float * logits = fh ? fh->dst : (float *) hm->data;
    const int64_t V = wh->ne[1];
    const bool sharded = shard_head && V >= 8 * (int64_t) N && mv_head_ok;
    struct hw { const void * wrows; const float * nw; int64_t v0, vc; };
    std::vector<hw> HW((size_t) N);
    for (int k = 0; k < (sharded ? N : 1); k++) {
        tp_rank & R = T.r[(size_t) k]; hw & h = HW[(size_t) k];
        h.v0 = 0; h.vc = V;
        if (sharded) cllm_tp_split(V, N, k, &h.v0, &h.vc);
        // this rank's rows of the lm_head: the host's tensor itself where the rank shares rank 0's GPU, else a copy made on first use
        h.wrows = (const char *) wh->data + (size_t) h.v0 * wh->nb[1]; h.nw = norm_w;
        if (R.gpu != c->device) {
            const ggml_tensor * src[1] = { wh };
            tp_piece pc = { h.wrows, (size_t) h.vc * wh->nb[1], (size_t) h.vc * wh->nb[1], 1, 0, (size_t) h.vc * wh->nb[1] };
            bool b2 = false;
            h.wrows = tp_get_shard(c, k, src, 1, &pc, 1, (size_t) h.vc * wh->nb[1], &b2);
            h.nw = replicate(k, norm_w, (size_t) d.H * 4, norm_t);
            if (!h.wrows || !h.nw) { HIPB_LOG("tensor parallel: the lm_head shard of rank %d could not be made (%s)", k, cllm_last_error()); cllm_set_device(c->device); return GGML_STATUS_FAILED; }
        }
    }
    if (!fh && !sharded) { cllm_tensor da = desc(hm->src[0]), db = desc(hm->src[1]); cllm_set_device(c->device); TPC(ensure_wdata(c, cllm_mul_mat_wsize(&da, &db))); }

    // ---- the launches of the step proper.  only == nullptr: every rank, site by site (ranks that share a stream must be issued in this order: a gather never polls for a launch
    //      behind it); only == a stream: the ranks of that stream alone, same order -- what is captured into that stream's graph ----
    auto issue = [&](void * only) -> ggml_status {
        auto mine = [&](int k) { return !only || T.r[(size_t) k].stream == only; };
        for (int k = 0; k < N; k++) if (mine(k)) {
            tp_rank & R = T.r[(size_t) k]; rk & x = X[(size_t) k];
            cllm_set_device(R.gpu);
            x.cur = x.xa;
            TPC(cllm_tp_fused_advance(R.fused, R.stream));
            if (table) TPC(cllm_op_rope_table(R.stream, x.pos, (int) d.hd, A0.freq_base, x.cs));
        }
        int pend = -1;                                // the site whose partial sums are not yet in the residual stream
        for (int l = 0; l < L; l++) {
            const fused_attn & A = P.attns[D.layers[l].attn];
            for (int k = 0; k < N; k++) if (mine(k)) {    // q | k | v (+ the all-reduce of the previous down projection + residual + RMS_NORM)
                tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k]; const ls & o = SH[(size_t) k][(size_t) l];
                cllm_set_device(R.gpu);
                cllm_tensor w = wdesc(tq, d.H, s.kvc * (d.gs + 2) * d.hd, o.qkv);
                if (pend < 0) TPC(cllm_op_mul_mat_vec_fused(R.stream, &w, 1, x.cur, o.an, W[(size_t) l].an_eps, 0, (const float *) o.bias, x.qkv));
                else { float * nx = x.cur == x.xa ? x.xb : x.xa; TPC(cllm_op_mul_mat_vec_tp_gather(R.stream, &w, x.cur, o.an, W[(size_t) l].an_eps, 0, (const float *) o.bias, x.qkv, R.fused, pend, nx)); x.cur = nx; }
            }
            for (int k = 0; k < N; k++) if (mine(k)) {    // RoPE + cache rows + attention over this rank's heads
                tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k];
                cllm_set_device(R.gpu);
                TPC(cllm_op_rope_kv_attn_decode(R.stream, x.qkv, x.pos, table ? x.cs : nullptr, A.freq_base, A.n_kv, (int)(s.kvc * d.gs), (int) s.kvc, (int) d.hd, A.mode, R.kv_mem[(size_t) 2 * l], R.kv_mem[(size_t) 2 * l + 1], d.ML,
                                                x.att, x.scores, x.score_bytes));
            }
            for (int k = 0; k < N; k++) if (mine(k)) {    // o: partial rows -> every rank
                tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k]; const ls & o = SH[(size_t) k][(size_t) l];
                cllm_set_device(R.gpu);
                cllm_tensor w = wdesc(to, s.kvc * d.gs * d.hd, d.H, o.o);
                TPC(cllm_op_mul_mat_vec_tp_scatter(R.stream, &w, 2, x.att, R.fused, 2 * l));
            }
            for (int k = 0; k < N; k++) if (mine(k)) {    // gate / up with SiLU * up (+ the all-reduce of o + residual + RMS_NORM)
                tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k]; const ls & o = SH[(size_t) k][(size_t) l];
                cllm_set_device(R.gpu);
                cllm_tensor w = wdesc(tg, d.H, 2 * s.fc, o.gu);
                float * nx = x.cur == x.xa ? x.xb : x.xa;
                TPC(cllm_op_mul_mat_vec_tp_gather(R.stream, &w, x.cur, o.fn, W[(size_t) l].fn_eps, 1, nullptr, x.act, R.fused, 2 * l, nx));
                x.cur = nx;
            }
            for (int k = 0; k < N; k++) if (mine(k)) {    // down: partial rows -> every rank
                tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k]; const ls & o = SH[(size_t) k][(size_t) l];
                cllm_set_device(R.gpu);
                cllm_tensor w = wdesc(td, s.fc, d.H, o.dn);
                TPC(cllm_op_mul_mat_vec_tp_scatter(R.stream, &w, 2, x.act, R.fused, 2 * l + 1));
            }
            pend = 2 * l + 1;
        }
        for (int k = 0; k < (sharded ? N : 1); k++) if (mine(k)) {      // the head
            tp_rank & R = T.r[(size_t) k]; rk & x = X[(size_t) k]; const hw & h = HW[(size_t) k];
            cllm_set_device(R.gpu);
            cllm_tensor w = wdesc(wh->type, d.H, h.vc, (void *) h.wrows);
            float * nx = x.cur == x.xa ? x.xb : x.xa;
            if (fh) {                             // a fused norm + lm_head: its RMS_NORM prologue takes the last all-reduce
                TPC(cllm_op_mul_mat_vec_tp_gather(R.stream, &w, x.cur, h.nw, eps_h, 0, nullptr, logits + h.v0, R.fused, pend, nx));
            } else {                              // chatllm's head: the last all-reduce lands in the residual stream (rank 0: the host's tensor), then norm + rows
                // (sharded: into the rank's own scratch -- ggml-alloc may have placed the logits on the freed block of the host's residual tensor, and the other ranks' rows land
                //  there while this rank's launch still reads its input; un-sharded: the host's residual tensor, as on one device)
                float * xfin = (k == 0 && !sharded) ? (float *) curx : nx;
                TPC(cllm_op_tp_gather_residual(R.stream, x.cur, d.H, R.fused, pend, xfin));
                if (k == 0) {
                    cllm_tensor dx = desc(hn->src[0]->src[0]), dwt = desc(hn->src[1]), dn = desc(hn);
                    dx.data = xfin;
                    dwt.ne[0] = hn->ne[0]; dwt.ne[1] = dwt.ne[2] = dwt.ne[3] = 1; dwt.nb[1] = dwt.nb[2] = dwt.nb[3] = (size_t) hn->ne[0] * 4;
                    TPC(cllm_op_rms_norm_mul(R.stream, &dx, &dwt, &dn, eps_h));
                }
                if (sharded) TPC(cllm_op_mul_mat_vec_fused(R.stream, &w, 1, xfin, h.nw, eps_h, 0, nullptr, logits + h.v0));      // (norm + quantize in the prologue: the bits of the two nodes)
                else {
                    cllm_tensor da = desc(hm->src[0]), db = desc(hm->src[1]), dd = desc(hm);
                    TPC(cllm_op_mul_mat(R.stream, &da, &db, &dd, c->wdata, c->wsize));
                }
            }
        }
        for (int k = 0; k < N; k++) if (mine(k)) {    // this step's cache row back into the host's caches
            tp_rank & R = T.r[(size_t) k]; const tp_split & s = S[(size_t) k]; rk & x = X[(size_t) k];
            cllm_set_device(R.gpu);
            TPC(cllm_op_kv_shard_copy(R.stream, R.kv_table, L, (int)(s.kvc * d.hd), (int)(d.nkv * d.hd), (int)(s.kv0 * d.hd), d.ML, 0, 0, x.pos, 1));
        }
        return GGML_STATUS_SUCCESS;
    };
    // ---- launch-list replay per stream (as be_graph_compute does for one stream): everything a launch depends on goes into a signature; the second identical step is captured,
    //      one graph per distinct stream (ranks on distinct GPUs: one each; virtual ranks: one for all), and replayed from then on.  CLLM_HIP_TP_GRAPH=0: every launch issued ----
    static const bool tp_graph = !getenv("CLLM_HIP_TP_GRAPH") || atoi(getenv("CLLM_HIP_TP_GRAPH")) != 0;
    std::vector<uint64_t> sig;
    {
        auto put = [&](const void * p) { sig.push_back((uint64_t)(uintptr_t) p); };
        auto putf = [&](float f) { uint32_t u; memcpy(&u, &f, 4); sig.push_back(u); };
        sig.push_back((uint64_t) N); sig.push_back((uint64_t) L); sig.push_back((uint64_t) d.H); sig.push_back((uint64_t) d.F); sig.push_back((uint64_t) d.hd); sig.push_back((uint64_t) d.ML);
        sig.push_back((uint64_t) tq | (uint64_t) to << 8 | (uint64_t) tg << 16 | (uint64_t) td << 24 | (uint64_t) wh->type << 32); sig.push_back((uint64_t) A0.mode); putf(A0.freq_base); putf(eps_h);
        sig.push_back((uint64_t) sharded | (uint64_t)(fh != nullptr) << 1 | (uint64_t) table << 2);
        put(logits); put(curx); put(hn ? hn->data : nullptr); put(hn ? hn->src[0]->src[0]->data : nullptr); put(hm ? hm->src[0]->data : nullptr); put(hm ? hm->src[1]->data : nullptr); put(c->wdata);
        for (int k = 0; k < N; k++) {
            const tp_rank & R = T.r[(size_t) k]; const tp_split & s0 = S[(size_t) k]; const rk & x = X[(size_t) k];
            put(R.stream); put(R.fused); put(R.scratch); put(R.kv_table); sig.push_back((uint64_t) s0.kv0 << 32 | (uint64_t) s0.kvc); sig.push_back((uint64_t) s0.f0 << 32 | (uint64_t) s0.fc);
            sig.push_back((uint64_t)(x.score_bytes != 0));
            put(HW[(size_t) k].wrows); put(HW[(size_t) k].nw); sig.push_back((uint64_t) HW[(size_t) k].v0); sig.push_back((uint64_t) HW[(size_t) k].vc);
            for (int l = 0; l < L; l++) { const ls & o = SH[(size_t) k][(size_t) l]; put(o.qkv); put(o.bias); put(o.o); put(o.gu); put(o.dn); put(o.an); put(o.fn); put(R.kv_mem[(size_t) 2 * l]); put(R.kv_mem[(size_t) 2 * l + 1]); putf(W[(size_t) l].an_eps); putf(W[(size_t) l].fn_eps); }
        }
    }
    std::vector<void *> streams;                   // the distinct streams, rank order
    for (int k = 0; k < N; k++) { bool seen = false; for (void * q : streams) seen = seen || q == T.r[(size_t) k].stream; if (!seen) streams.push_back(T.r[(size_t) k].stream); }
    auto gpu_of = [&](void * q) { for (int k = 0; k < N; k++) if (T.r[(size_t) k].stream == q) return T.r[(size_t) k].gpu; return c->device; };
    bool replayed = false, ahead_hit = false;
    if (c->ahead.inflight) {                       // a sharded step is running ahead of the host (tp_ahead_replay): is it this one?
        const int hr = ahead_resolve(c, c->ahead.tp && T.rp.valid && !T.execs.empty() && sig == T.graph_sig && T.rp.n_kv == A0.n_kv, cur_sets);
        if (hr < 0) return GGML_STATUS_FAILED;
        ahead_hit = hr == 1;
    }
    if (ahead_hit) { replayed = true; T.replays++; }
    else if (const ggml_status ps = pre_step(); ps != GGML_STATUS_SUCCESS) return ps;
    if (ahead_hit) {}
    else if (tp_graph && !T.graph_broken && T.execs.size() == streams.size() && !T.execs.empty() && sig == T.graph_sig) {
        for (size_t q = 0; q < streams.size(); q++) { cllm_set_device(gpu_of(streams[q])); TPC(cllm_graph_launch(T.execs[q], streams[q])); }
        replayed = true; T.replays++;
    } else if (tp_graph && !T.graph_broken && sig == T.last_sig) {      // second time in a row: capture one graph per stream, then launch them
        if (!T.execs.empty()) { for (void * q : streams) { cllm_set_device(gpu_of(q)); cllm_stream_sync(q); } }      // (a replaced graph may still be running)
        for (void * e : T.execs) if (e) cllm_graph_destroy(e);
        T.execs.clear(); T.graph_sig.clear();
        bool ok = true;
        for (size_t q = 0; q < streams.size() && ok; q++) {
            cllm_set_device(gpu_of(streams[q]));
            if (cllm_graph_capture_begin(streams[q]) != CLLM_OK) { ok = false; break; }
            const ggml_status rs = issue(streams[q]);
            void * exec = nullptr;
            cllm_set_device(gpu_of(streams[q]));
            const int rc = cllm_graph_capture_end(streams[q], &exec);
            if (rs != GGML_STATUS_SUCCESS || rc != CLLM_OK || !exec) { if (exec) cllm_graph_destroy(exec); ok = false; break; }
            T.execs.push_back(exec);
        }
        if (!ok) {                                 // nothing ran (captures execute nothing): issue the step the plain way, and stop trying
            HIPB_LOG("tensor parallel: launch-list capture failed (%s): issuing the launches of every step from now on", cllm_last_error());
            for (void * e : T.execs) if (e) cllm_graph_destroy(e);
            T.execs.clear(); T.graph_broken = true;
            const ggml_status rs = issue(nullptr);
            if (rs != GGML_STATUS_SUCCESS) return rs;
        } else {
            T.graph_sig = sig; T.captures++;
            for (size_t q = 0; q < streams.size(); q++) { cllm_set_device(gpu_of(streams[q])); TPC(cllm_graph_launch(T.execs[q], streams[q])); }
            replayed = true;
        }
    } else {
        const ggml_status rs = issue(nullptr);
        if (rs != GGML_STATUS_SUCCESS) return rs;
    }
    T.last_sig.swap(sig);
    if (!ahead_hit) for (int k = 0; k < N; k++) {   // everything joins rank 0's stream
        tp_rank & R = T.r[(size_t) k];
        if (R.own_stream) { cllm_set_device(R.gpu); TPC(cllm_event_record(R.ev, R.stream)); cllm_set_device(c->device); TPC(cllm_stream_wait_event(st0, R.ev)); }
    }
    {   // what a step started ahead of the host will need (tp_ahead_replay): valid while the captured graphs are
        auto & Q = T.rp;
        Q.valid = replayed && !T.execs.empty();
        if (Q.valid) {
            Q.ea = desc(emb->src[0]); Q.eb = desc(emb->src[1]); Q.ed = desc(emb); Q.pos_src = A0.pos; Q.hbytes = (size_t) d.H * 4; Q.n_kv = A0.n_kv;
            Q.xa.clear(); Q.pos.clear();
            for (int k = 0; k < N; k++) { Q.xa.push_back(X[(size_t) k].xa); Q.pos.push_back(X[(size_t) k].pos); }
            Q.streams = streams; Q.sgpu.clear();
            for (void * q : streams) Q.sgpu.push_back(gpu_of(q));
        }
    }
    *replayed_out = replayed;
#undef TPC
#undef TP_NO
    cllm_set_device(c->device);
    T.steps++; T.last_tp = true;
    if (g_stats) HIPB_LOG("HIP0 graph_compute: %d nodes -> tensor parallel over %d ranks: %d launches per rank (%d layers x 5 + head%s)%s, all-reduce fused into the mat-vecs", ggml_graph_n_nodes(g), N, 5 * L + 6, L,
                          sharded ? ", lm_head rows sharded" : "", ahead_hit ? ", started ahead of the host" : replayed ? ", replayed from the captured graphs" : "");
    return GGML_STATUS_SUCCESS;
}

// the NEXT sharded step, started ahead of the host (ahead_launch): the token id and the positions are already on rank 0 (the prep launch in front of this wrote them); what
// tp_graph_compute's pre_step does -- embedding row, hand-over to the ranks, events -- then the captured graphs and the joins.  The cache shards need no refresh: the previous
// step left them one row short of this one, and this step writes that row.
int tp_ahead_replay(hip_backend_ctx * c) {
    auto & T = g_tp; auto & Q = T.rp;
    if (!Q.valid || T.execs.empty() || T.execs.size() != Q.streams.size()) return CLLM_E_INVALID;
    void * st0 = c->stream;
    const int N = T.n;
#define TRY_(expr) do { const int rc_ = (expr); if (rc_ != CLLM_OK) { cllm_set_device(c->device); return rc_; } } while (0)
    cllm_set_device(c->device);
    TRY_(cllm_op_get_rows(st0, &Q.ea, &Q.eb, &Q.ed));
    bool remote = false;
    for (int k = 0; k < N; k++) {
        tp_rank & R = T.r[(size_t) k];
        TRY_(cllm_memcpy_peer_async(Q.xa[(size_t) k], R.gpu, Q.ed.data, c->device, Q.hbytes, st0));
        TRY_(cllm_memcpy_peer_async(Q.pos[(size_t) k], R.gpu, Q.pos_src, c->device, 4, st0));
        remote = remote || R.own_stream;
    }
    if (remote) {
        TRY_(cllm_event_record(T.ev0, st0));
        for (int k = 0; k < N; k++) if (T.r[(size_t) k].own_stream) { cllm_set_device(T.r[(size_t) k].gpu); TRY_(cllm_stream_wait_event(T.r[(size_t) k].stream, T.ev0)); }
    }
    for (size_t q = 0; q < Q.streams.size(); q++) { cllm_set_device(Q.sgpu[q]); TRY_(cllm_graph_launch(T.execs[q], Q.streams[q])); }
    for (int k = 0; k < N; k++) {
        tp_rank & R = T.r[(size_t) k];
        if (R.own_stream) { cllm_set_device(R.gpu); TRY_(cllm_event_record(R.ev, R.stream)); cllm_set_device(c->device); TRY_(cllm_stream_wait_event(st0, R.ev)); }
        R.kv_valid = Q.n_kv + 1;
    }
#undef TRY_
    cllm_set_device(c->device);
    Q.n_kv += 1; T.steps++; T.last_tp = true;
    return CLLM_OK;
}

// arm decode-ahead for the token after this one (see ahead_launch) -- and, with the chain on, start that step right away.  have_graph: this step was replayed from a captured
// launch list (one graph, or the tensor-parallel device's graph per stream) that the next step can be started with.
void ahead_arm(hip_backend_ctx * c, ggml_cgraph * g, const fuse_plan & plan, const std::vector<hip_backend_ctx::scalar_set> & cur_sets, bool have_graph, bool tp) {
    static const bool ahead_off = getenv("CLLM_HIP_AHEAD") && atoi(getenv("CLLM_HIP_AHEAD")) == 0;
    auto & A = c->ahead;
    A.armed = false;
    const int nn = ggml_graph_n_nodes(g);
    if (ahead_off || !have_graph || A.graphs < A.skip_until || cur_sets.empty() || nn <= 0 || plan.attns.empty() || c->device >= 64) return;
    const ggml_tensor * out = ggml_graph_node(g, nn - 1), * ids = nullptr;
    int n_out = 0; A.outs.clear(); bool ok = out->type == GGML_TYPE_F32 && ggml_is_contiguous(out) && out->data && ggml_nbytes(out) % 4 == 0 && ggml_nbytes(out) >= 8;
    int64_t min_ml = INT64_MAX;
    for (const fused_attn & F : plan.attns) { ok = ok && F.level == 2; if (F.ML < min_ml) min_ml = F.ML; }
    for (int i = 0; ok && i < nn; i++) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        if (t->flags & GGML_TENSOR_FLAG_OUTPUT) { n_out++; if (t->data && ggml_is_contiguous(t)) A.outs.push_back({ (const char *) t->data, ggml_nbytes(t), 0 }); else ok = false; }
        if (!plan.skip[i] && (t->op == GGML_OP_SET_ROWS || t->op == GGML_OP_FLASH_ATTN_EXT)) ok = false;      // cache writes outside the fused block, growing masks: not predicted (experts are picked on the device: nothing to predict)
        if (!ids && t->op == GGML_OP_GET_ROWS && t->src[1] && t->src[1]->type == GGML_TYPE_I32 && ggml_nelements(t->src[1]) == 1) ids = t->src[1];
    }
    if (!(out->flags & GGML_TENSOR_FLAG_OUTPUT)) A.outs.push_back({ (const char *) out->data, ggml_nbytes(out), 0 });
    ok = ok && ids && ids->data && n_out <= 8;
    bool has_ids = false;
    for (const auto & ss : cur_sets) { if (ids && ss.ptr == ids->data) has_ids = true; else if ((int64_t) ss.val + 1 >= min_ml || ss.val < 0) ok = false; }
    static const bool dbg = getenv("CLLM_HIP_AHEAD_DEBUG") != nullptr;
    if (dbg) HIPB_LOG("ahead: ok=%d has_ids=%d ids=%p n_out=%d out_flag=%d min_ml=%lld sets=%zu out=%s(%s) bytes=%zu", (int) ok, (int) has_ids, ids ? ids->data : nullptr, n_out, (int)((out->flags & GGML_TENSOR_FLAG_OUTPUT) != 0), (long long) min_ml, cur_sets.size(), out->name, ggml_op_name(out->op), ggml_nbytes(out));
    if (!ok || !has_ids) return;
    A.ids_ptr = ids->data; A.logits_ptr = out->data; A.logits_bytes = ggml_nbytes(out); A.last_sets = cur_sets;
    A.armed = true; A.tp = tp; g_ahead_ctx[c->device] = c;
    // the next step goes out NOW, behind the one just launched (CLLM_HIP_AHEAD_CHAIN=0: when the host reads the logits, round 5's order)
    static const bool chain = !getenv("CLLM_HIP_AHEAD_CHAIN") || atoi(getenv("CLLM_HIP_AHEAD_CHAIN")) != 0;
    if (chain) ahead_launch(c, true);
}

ggml_status be_graph_compute(ggml_backend_t backend, ggml_cgraph * g) {
    auto * c = (hip_backend_ctx *) backend->context;
    cllm_set_device(c->device);
    if (c->kernel_error) return GGML_STATUS_FAILED;            // (an earlier launch's in-kernel wait timed out: be_sync)
    void * st = c->stream;
    static const bool trace = getenv("CLLM_HIP_TRACE") != nullptr;
    if (trace) {
        HIPB_LOG("graph_compute: %d nodes", ggml_graph_n_nodes(g));
        for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
            const ggml_tensor * n = ggml_graph_node(g, i);
            fprintf(stderr, "  %3d %-14s %-24s [%lld,%lld,%lld,%lld] nb0=%zu", i, ggml_op_name(n->op), n->name, (long long) n->ne[0], (long long) n->ne[1], (long long) n->ne[2], (long long) n->ne[3], n->nb[0]);
            for (int k = 0; k < 3; k++) if (n->src[k]) fprintf(stderr, "  s%d=%s(%s)[%lld,%lld,%lld]", k, n->src[k]->name, ggml_op_name(n->src[k]->op), (long long) n->src[k]->ne[0], (long long) n->src[k]->ne[1], (long long) n->src[k]->ne[2]);
            fputc('\n', stderr);
        }
    }
    ws_scope ws(g_ws.issue_us);
    flush_sets();                               // queued small set_tensor copies (null stream; this stream does not wait for it by itself)
    cllm_set_device(c->device);
    std::vector<hip_backend_ctx::scalar_set> cur_sets;
    { std::lock_guard<std::mutex> lock(g_ring.m); if (c->device < 64) cur_sets.swap(g_scalar_sets[c->device]); }
    if (c->device < 64 && g_ahead_ctx[c->device] && g_ahead_ctx[c->device] != c && g_ahead_ctx[c->device]->ahead.inflight) {
        // another backend context of this device (a draft / embedding / accessory model) has a step running ahead: buf_set held this graph's scalar writes back
        // for THAT context's hit test -- they are ours.  Let the step finish, put them in place, and leave that context to run its next step the normal way.
        hip_backend_ctx * o = g_ahead_ctx[c->device];
        ahead_quiesce(c->device);
        o->ahead.inflight = false; o->ahead.armed = false; o->ahead.misses = 0;
        cllm_set_device(c->device);
        for (const auto & ss : cur_sets) if (cllm_memcpy_h2d((void *) ss.ptr, &ss.val, 4, nullptr) != CLLM_OK) { HIPB_LOG("scalar write failed: %s", cllm_last_error()); return GGML_STATUS_FAILED; }
        if (cllm_stream_sync(nullptr) != CLLM_OK) return GGML_STATUS_FAILED;
    }
    c->ahead.graphs++;
    fuse_plan plan = make_plan(g);
    if (g_tp.n > 1) {
        // the logical tensor-parallel device: a recognised decode step runs sharded over the ranks (tp_graph_compute); anything else runs below, un-sharded on rank 0 -- and may
        // write the KV caches behind the shards' back
        bool tp_replayed = false;
        const ggml_status ts = tp_graph_compute(c, g, plan, cur_sets, &tp_replayed);
        if (ts != GGML_STATUS_ABORTED) {
            c->last_sig.clear();
            if (ts == GGML_STATUS_SUCCESS) ahead_arm(c, g, plan, cur_sets, tp_replayed, true);      // the next sharded step may start ahead of the host, as on one device
            return ts;
        }
        if (c->ahead.inflight && c->ahead.tp) {       // a sharded step runs ahead and the host asks for something else (a prompt, another model): let it finish, put the host's scalars in place
            if (ahead_resolve(c, false, cur_sets) < 0) return GGML_STATUS_FAILED;
        }
        g_tp_kv_epoch++; g_tp.plain++; g_tp.last_tp = false; g_tp.rp.valid = false;
    }
    if (trace) {
        for (const fused_mv & f : plan.mvs) if (f.node >= 0) fprintf(stderr, "  plan: mat-vec node %d pro %d%s%s%s\n", f.node, f.pro, f.resid ? " +resid" : "", f.alias ? " STAGED (dst overlaps an input)" : "", f.group >= 0 ? " grouped" : "");
        for (const fused_attn & A : plan.attns) fprintf(stderr, "  plan: attention level %d%s\n", A.level, A.alias ? " STAGED" : "");
        for (int i = 0; i < ggml_graph_n_nodes(g); i++) if (!plan.skip[i]) { const ggml_tensor * n = ggml_graph_node(g, i); if (n->op == GGML_OP_ADD || n->op == GGML_OP_MUL || n->op == GGML_OP_CPY || n->op == GGML_OP_CONT) fprintf(stderr, "  plan: node %d %s launched on its own (alt %d)\n", i, ggml_op_name(n->op), plan.alt[i]); }
    }
    if (g_stats) g_ws.plan_us += wall_stats::us(ws.t0, wall_stats::clk::now());
    // fused attention: [cos/sin table 1 KB][q | k | v projections][scores of the long-context form]
    // scratch of the fused forms: [cos/sin table 1 KB][q | k | v projections][SiLU(gate)*up activation][scores of the long-context attention]
    size_t qkv_bytes = 0, act_bytes = 0, score_bytes = 0;
    for (const fused_attn & A : plan.attns) if (A.level == 2) {
        const size_t q = ((size_t) A.hd * (A.nh + 2 * A.nkv) * 4 + 255) & ~(size_t) 255, sb = cllm_attn_decode_wsize(A.n_kv, A.nh, A.ML);
        if (q > qkv_bytes) qkv_bytes = q;
        if (sb > score_bytes) score_bytes = sb;
    }
    for (const merge_group & G : plan.groups) if (G.interleave) {
        const size_t a = ((size_t) ggml_graph_node(g, plan.mvs[G.member[0]].node)->ne[0] * 4 + 255) & ~(size_t) 255;
        if (a > act_bytes) act_bytes = a;
    }
    size_t stage_bytes = 0;              // outputs of fused launches whose dst aliases one of their inputs (see make_plan)
    for (const fused_mv & f : plan.mvs) if (f.alias) { const size_t b = ((size_t) ggml_graph_node(g, f.node)->src[0]->ne[1] * 4 + 255) & ~(size_t) 255; if (b > stage_bytes) stage_bytes = b; }
    for (const fused_attn & A : plan.attns) if (A.alias) { const size_t b = ((size_t) A.nh * A.hd * 4 + 255) & ~(size_t) 255; if (b > stage_bytes) stage_bytes = b; }
    for (const fused_fa & F : plan.fas) if (F.alias) { const size_t b = (F.bytes + 255) & ~(size_t) 255; if (b > stage_bytes) stage_bytes = b; }
    float * a_cs = nullptr, * a_qkv = nullptr, * a_act = nullptr, * a_stage = nullptr; void * a_scores = nullptr;
    if (qkv_bytes || act_bytes || stage_bytes) {
        if (int rc = ensure_abuf(c, 1024 + qkv_bytes + act_bytes + score_bytes + stage_bytes)) { HIPB_LOG("fusion scratch: %s", cllm_last_error()); return rc == CLLM_E_ALLOC ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_FAILED; }
        a_cs = (float *) c->abuf; a_qkv = (float *)((char *) c->abuf + 1024); a_act = (float *)((char *) c->abuf + 1024 + qkv_bytes);
        a_scores = (char *) c->abuf + 1024 + qkv_bytes + act_bytes;
        a_stage = (float *)((char *) c->abuf + 1024 + qkv_bytes + act_bytes + score_bytes);
        for (const fused_attn & A : plan.attns) if (A.level == 2) {
            plan.mvs[A.wq].dst = a_qkv; plan.mvs[A.wk].dst = a_qkv + (size_t) A.hd * A.nh; plan.mvs[A.wv].dst = a_qkv + (size_t) A.hd * (A.nh + A.nkv);
        }
    }
    // merged expert gate/up launches: only where a packed (per-expert interleaved) copy of the two weight tensors can be had
    for (moe_gate_up & G : plan.gus) {
        const ggml_tensor * w[2] = { ggml_graph_node(g, G.gate)->src[0], ggml_graph_node(g, G.up)->src[0] };
        G.W = get_pack(c->device, st, w, 2, true);
        if (!G.W) continue;
        plan.skip[G.gate] = plan.skip[G.up] = 1;            // (the UNARY is skipped already)
        plan.alt[G.mul] = ALT_MOE_GATE_UP; plan.moe[G.mul] = (int)(&G - plan.gus.data());
        // the block's router into the same launch (one launch less per sparse-MoE block): the experts read the ids of a router launch whose normalised activation feeds nothing but
        // that router and these two MUL_MAT_IDs, router and experts share one weight type, and whatever else reads the probabilities / ids runs after this node (or is itself fused away)
        static const bool fold = !getenv("CLLM_HIP_MOE_FOLD") || atoi(getenv("CLLM_HIP_MOE_FOLD")) != 0;
        const ggml_tensor * gm = ggml_graph_node(g, G.gate);
        for (size_t ri = 0; fold && ri < plan.routers.size(); ri++) {
            moe_router & R = plan.routers[ri];
            if (R.itk < 0 || plan.skip[R.itk] || plan.alt[R.itk] != ALT_MOE_ROUTER || ggml_graph_node(g, R.itk) != gm->src[2] || R.gate->type != gm->src[0]->type) continue;
            const ggml_tensor * act = gm->src[1];
            while (act != R.xnorm && (act->op == GGML_OP_RESHAPE || act->op == GGML_OP_VIEW) && act->src[0] && act->src[0]->data == act->data) act = act->src[0];
            if (act != R.xnorm || ggml_nelements(gm->src[1]) != R.xnorm->ne[0]) continue;
            bool ok = R.xn_users.size() == 2 && ((R.xn_users[0] == G.gate && R.xn_users[1] == G.up) || (R.xn_users[0] == G.up && R.xn_users[1] == G.gate));
            for (int u : R.out_users) if (u != G.gate && u != G.up && !plan.skip[u] && u <= G.mul) ok = false;
            // the fold moves the router from the TOP_K node to the later MUL node: the launch there still READS R.x, and its fallback (two launches) WRITES R.xnorm -- blocks
            // ggml-alloc may have handed to a node in between (their last readers precede the MUL).  Nothing launched in (TOP_K, MUL] may overlap either.
            for (int j = R.itk + 1; ok && j <= G.mul; j++) {
                const ggml_tensor * t = ggml_graph_node(g, j);
                if (plan.skip[j] || !t->data || t->op == GGML_OP_NONE || t->op == GGML_OP_RESHAPE || t->op == GGML_OP_VIEW || t->op == GGML_OP_PERMUTE || t->op == GGML_OP_TRANSPOSE) continue;
                const size_t kb = (size_t) R.xnorm->ne[0] * 4;
                if (overlap(t->data, ggml_nbytes(t), R.x->data, kb) || overlap(t->data, ggml_nbytes(t), R.xnorm->data, kb)) ok = false;
            }
            if (!ok) continue;
            G.router = (int) ri; plan.skip[R.itk] = 1;
            break;
        }
    }
    int merged = 0, launches = 0;
    // one walk over the nodes; sw != nullptr: serialize the calls instead of making them (same decisions, same host-side state changes)
    auto walk = [&](fuse_plan & plan, sig_writer * sw) -> ggml_status {
#define CALL(fn, ...) (sw ? (sw->call((const void *) fn, __VA_ARGS__), (int) CLLM_OK) : fn(__VA_ARGS__))
    merged = launches = 0;
    const int32_t * tab_pos = nullptr; int tab_hd = 0; float tab_fb = 0.0f;      // what a_cs holds (computed once per graph, not per layer)
    for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
        ggml_tensor * n = ggml_graph_node(g, i);
        if (ggml_is_empty(n) || plan.skip[i]) continue;
        launches += !(n->op == GGML_OP_NONE || n->op == GGML_OP_RESHAPE || n->op == GGML_OP_VIEW || n->op == GGML_OP_PERMUTE || n->op == GGML_OP_TRANSPOSE);
        const ggml_tensor * a = n->src[0], * b = n->src[1];
        int rc = CLLM_OK;
        cllm_tensor d = desc(n), da, db, dc;
        if (a) da = desc(a);
        if (b) db = desc(b);
        switch (n->op) {
            case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_MUL_MAT: if (plan.alt[i] == ALT_FLASH_PREFILL) {
                const fused_fa & F = plan.fas[plan.moe[i]];
                const ggml_tensor * kq = ggml_graph_node(g, F.ikq);
                cllm_tensor dk = desc(kq->src[0]), dq = desc(kq->src[1]);
                if (F.alias) {
                    cllm_tensor ds = d; ds.data = a_stage;
                    rc = CALL(cllm_op_attn_prefill, st, &dq, &dk, &da, &ds, F.scale, F.n_past);
                    if (rc == CLLM_OK) rc = CALL(cllm_memcpy_d2d, (void *) n->data, (const void *) a_stage, F.bytes, st);
                } else rc = CALL(cllm_op_attn_prefill, st, &dq, &dk, &da, &d, F.scale, F.n_past);
            } else if (plan.mv[i] >= 0) {
                const fused_mv & f = plan.mvs[plan.mv[i]];
                if (f.group >= 0) {
                    merge_group & G = plan.groups[f.group];
                    if (G.state == 0) {         // the first member in node order launches for all of them -- if a packed copy of the weights can be had
                        const ggml_tensor * w[3]; int64_t rows = 0;
                        for (int k = 0; k < G.n; k++) { w[k] = ggml_graph_node(g, plan.mvs[G.member[k]].node)->src[0]; rows += w[k]->ne[1]; }
                        void * W = get_pack(c->device, st, w, G.n, G.interleave);
                        const float * Bv = nullptr;
                        if (W && G.bias) {
                            const ggml_tensor * bt[3];
                            for (int k = 0; k < G.n; k++) bt[k] = plan.mvs[G.member[k]].resid_t;
                            Bv = (const float *) get_pack(c->device, st, bt, G.n, false, true);
                            if (!Bv) W = nullptr;
                        }
                        G.state = 2;
                        if (W) {
                            cllm_tensor dw = desc(w[0]); dw.ne[1] = rows; dw.nb[2] = dw.nb[3] = (size_t) rows * dw.nb[1]; dw.data = W;
                            rc = CALL(cllm_op_mul_mat_vec_fused, st, &dw, 1, f.px, f.pw, f.eps, G.interleave ? 1 : 0, Bv, G.interleave ? a_act : plan.mvs[G.member[0]].dst);
                            if (rc == CLLM_OK) {
                                G.state = 1; merged++;
                                if (G.consumer >= 0) { fused_mv & d = plan.mvs[G.consumer]; d.pro = 2; d.px = a_act; d.pw = nullptr; }
                            } else if (rc == CLLM_E_UNSUPPORTED) rc = CLLM_OK;      // shape the merged form does not take: launch the members
                        }
                    }
                    if (G.state == 1 || rc != CLLM_OK) break;
                }
                if (f.alias) {       // staged: dst overlaps px / pw (with an in-place residual the staged launch still reads resid == the final dst: no hazard)
                    rc = CALL(cllm_op_mul_mat_vec_fused, st, &da, f.pro, f.px, f.pw, f.eps, 0, f.resid, a_stage);
                    if (rc == CLLM_OK) rc = CALL(cllm_memcpy_d2d, (void *) f.dst, (const void *) a_stage, (size_t) a->ne[1] * 4, st);
                } else rc = CALL(cllm_op_mul_mat_vec_fused, st, &da, f.pro, f.px, f.pw, f.eps, 0, f.resid, f.dst);
            } else if (plan.pf[i] >= 0) {
                const pf_mm & F = plan.pfs[plan.pf[i]];
                const size_t need = cllm_mul_mat_wsize(&da, &db);
                if ((rc = ensure_wdata(c, need))) break;
                cllm_tensor dx = F.pro == 5 ? db : desc(F.x), dw2, dr, dout = F.out ? desc(F.out) : d;
                if (F.pro == 1) { dx.ne[0] = db.ne[0]; dx.ne[1] = db.ne[1]; dx.ne[2] = dx.ne[3] = 1; dx.nb[1] = (size_t) db.ne[0] * 4; dx.nb[2] = dx.nb[3] = dx.nb[1] * (size_t) db.ne[1]; }
                if (F.w2) { dw2 = desc(F.w2); if (F.pro == 1) { dw2.ne[0] = db.ne[0]; dw2.ne[1] = dw2.ne[2] = dw2.ne[3] = 1; dw2.nb[1] = dw2.nb[2] = dw2.nb[3] = (size_t) db.ne[0] * 4; } }
                if (F.resid) dr = desc(F.resid);
                rc = CALL(cllm_op_mul_mat_ex, st, &da, &dx, &dout, c->wdata, c->wsize, F.pro, F.w2 ? &dw2 : nullptr, F.eps, 0, F.resid ? &dr : nullptr);
            } else {
                const size_t need = cllm_mul_mat_wsize(&da, &db);
                if ((rc = ensure_wdata(c, need))) break;
                rc = CALL(cllm_op_mul_mat, st, &da, &db, &d, c->wdata, c->wsize);
            } break;
            case GGML_OP_MUL_MAT_ID: {
                dc = desc(n->src[2]);
                const size_t need = cllm_mul_mat_wsize(&da, &db);
                if ((rc = ensure_wdata(c, need))) break;
                rc = CALL(cllm_op_mul_mat_id, st, &da, &db, &dc, &d, c->wdata, c->wsize);
            } break;
            case GGML_OP_RMS_NORM: { float eps; memcpy(&eps, n->op_params, 4); rc = CALL(cllm_op_rms_norm, st, &da, &d, eps); } break;
            case GGML_OP_ADD: if (plan.alt[i] == ALT_MOE_COMBINE) {
                const fused_moe & M = plan.moes[plan.moe[i]];
                cllm_tensor de = desc(M.experts), dp = desc(M.probs), di = desc(M.ids), dr;
                dp.ne[0] = M.probs->ne[0]; dp.ne[1] = M.experts->ne[2]; dp.ne[2] = dp.ne[3] = 1; dp.nb[1] = (size_t) dp.ne[0] * 4; dp.nb[2] = dp.nb[3] = dp.nb[1] * (size_t) dp.ne[1];
                if (M.resid) { dr = desc(M.resid); dr.ne[0] = M.experts->ne[0]; dr.ne[1] = M.experts->ne[2]; dr.ne[2] = dr.ne[3] = 1; dr.nb[1] = (size_t) dr.ne[0] * 4; dr.nb[2] = dr.nb[3] = dr.nb[1] * (size_t) dr.ne[1]; }
                cllm_tensor dd = d; dd.ne[0] = M.experts->ne[0]; dd.ne[1] = M.experts->ne[2]; dd.ne[2] = dd.ne[3] = 1; dd.nb[1] = (size_t) dd.ne[0] * 4; dd.nb[2] = dd.nb[3] = dd.nb[1] * (size_t) dd.ne[1];
                if (M.down >= 0) {
                    const ggml_tensor * mm = ggml_graph_node(g, M.down);
                    cllm_tensor dw = desc(mm->src[0]), dx = desc(mm->src[1]);
                    rc = CALL(cllm_op_mul_mat_id_combine, st, &dw, &dx, &di, &dp, M.resid ? &dr : nullptr, &dd);
                } else rc = CALL(cllm_op_moe_combine, st, &de, &dp, &di, M.resid ? &dr : nullptr, &dd);
            } else rc = CALL(cllm_op_add, st, &da, &db, &d); break;
            case GGML_OP_MUL: if (plan.alt[i] == ALT_MOE_GATE_UP) {
                const moe_gate_up & G = plan.gus[plan.moe[i]];
                const ggml_tensor * gm = ggml_graph_node(g, G.gate);
                cllm_tensor dw = desc(gm->src[0]), dx = desc(gm->src[1]), di = desc(gm->src[2]);
                dw.ne[1] *= 2; dw.nb[2] *= 2; dw.nb[3] = dw.nb[2] * (size_t) dw.ne[2]; dw.data = G.W;
                if (G.router >= 0) {
                    const moe_router & R = plan.routers[G.router];
                    cllm_tensor rx = desc(R.x), rw = desc(R.w), rg = desc(R.gate), rn = desc(R.xnorm), rp = desc(R.probs), rt = desc(ggml_graph_node(g, R.itk));
                    const int64_t K = R.xnorm->ne[0];
                    for (cllm_tensor * v : { &rx, &rw, &rn }) { v->ne[0] = K; v->ne[1] = v->ne[2] = v->ne[3] = 1; v->nb[1] = v->nb[2] = v->nb[3] = (size_t) K * 4; }
                    rc = CALL(cllm_op_moe_router_gate_up, st, &rx, &rw, R.eps, &rg, &dw, &rp, &rt, &d);
                    if (rc == CLLM_E_UNSUPPORTED) {              // the two launches it replaces
                        rc = CALL(cllm_op_moe_router, st, &rx, &rw, R.eps, &rg, &rn, &rp, &rt);
                        if (rc == CLLM_OK) rc = CALL(cllm_op_mul_mat_id_silu_mul, st, &dw, &dx, &di, &d);
                    }
                } else
                rc = CALL(cllm_op_mul_mat_id_silu_mul, st, &dw, &dx, &di, &d);
            } else if (plan.alt[i] == ALT_SILU_MUL || plan.alt[i] == ALT_MUL_SILU) {
                const bool a_is_silu = plan.alt[i] == ALT_SILU_MUL;
                const ggml_tensor * gate = (a_is_silu ? a : b)->src[0], * up = a_is_silu ? b : a;
                cllm_tensor dg = desc(gate), du = desc(up);
                rc = CALL(cllm_op_silu_mul, st, &dg, &du, &d);
            } else if (plan.alt[i] == ALT_RMS_NORM_MUL) {
                float eps; memcpy(&eps, a->op_params, 4);
                cllm_tensor dx = desc(a->src[0]), dwt = desc(b);
                dwt.ne[0] = n->ne[0]; dwt.ne[1] = dwt.ne[2] = dwt.ne[3] = 1; dwt.nb[1] = dwt.nb[2] = dwt.nb[3] = (size_t) n->ne[0] * 4;
                rc = CALL(cllm_op_rms_norm_mul, st, &dx, &dwt, &d, eps);
            } else rc = CALL(cllm_op_mul, st, &da, &db, &d); break;
            case GGML_OP_DIV: rc = CALL(cllm_op_div, st, &da, &db, &d); break;
            case GGML_OP_SUM_ROWS: rc = CALL(cllm_op_sum_rows, st, &da, &d); break;
            case GGML_OP_TOP_K: if (plan.alt[i] == ALT_MOE_ROUTER) {
                const moe_router & R = plan.routers[plan.moe[i]];
                cllm_tensor dx = desc(R.x), dwt = desc(R.w), dg = desc(R.gate), dn = desc(R.xnorm), dp = desc(R.probs);
                const int64_t K = R.xnorm->ne[0];
                for (cllm_tensor * v : { &dx, &dwt, &dn }) { v->ne[0] = K; v->ne[1] = v->ne[2] = v->ne[3] = 1; v->nb[1] = v->nb[2] = v->nb[3] = (size_t) K * 4; }
                rc = CALL(cllm_op_moe_router, st, &dx, &dwt, R.eps, &dg, &dn, &dp, &d);
            } else rc = CALL(cllm_op_top_k, st, &da, &d); break;
            case GGML_OP_SCALE: { float s, bias; memcpy(&s, n->op_params, 4); memcpy(&bias, (const float *) n->op_params + 1, 4); rc = CALL(cllm_op_scale, st, &da, &d, s, bias); } break;
            case GGML_OP_DIAG_MASK_INF: rc = CALL(cllm_op_diag_mask_inf, st, &da, &d, n->op_params[0]); break;
            case GGML_OP_UNARY: rc = CALL(cllm_op_unary, st, CLLM_UNARY_SILU, &da, &d); break;
            case GGML_OP_ROPE: {
                cllm_rope_params p;
                p.n_dims = n->op_params[1]; p.mode = n->op_params[2]; p.n_ctx_orig = n->op_params[4];
                memcpy(&p.freq_base, n->op_params + 5, 4); memcpy(&p.freq_scale, n->op_params + 6, 4); memcpy(&p.ext_factor, n->op_params + 7, 4);
                memcpy(&p.attn_factor, n->op_params + 8, 4); memcpy(&p.beta_fast, n->op_params + 9, 4); memcpy(&p.beta_slow, n->op_params + 10, 4);
                if (n->src[2]) dc = desc(n->src[2]);
                rc = CALL(cllm_op_rope, st, &da, &db, n->src[2] ? &dc : nullptr, &d, &p);
            } break;
            case GGML_OP_SOFT_MAX: if (plan.sm_src[i] >= 0) {
                const ggml_tensor * sc = ggml_graph_node(g, plan.sm_src[i]);
                float scale; memcpy(&scale, sc->op_params, 4);
                cllm_tensor ds = desc(sc->src[0]);
                rc = CALL(cllm_op_scale_mask_soft_max, st, &ds, &d, scale, n->src[0]->op_params[0]);
            } else {
                float scale, max_bias; memcpy(&scale, n->op_params, 4); memcpy(&max_bias, (const float *) n->op_params + 1, 4);
                rc = CALL(cllm_op_soft_max, st, &da, b ? &db : nullptr, &d, scale, max_bias);
            } break;
            case GGML_OP_SET_ROWS: rc = CALL(cllm_op_set_rows, st, &da, &db, &d); break;          // dst is a view of the cache (node->data)
            case GGML_OP_FLASH_ATTN_EXT: {
                float scale, max_bias, softcap; memcpy(&scale, n->op_params, 4); memcpy(&max_bias, (const float *) n->op_params + 1, 4); memcpy(&softcap, (const float *) n->op_params + 2, 4);
                dc = desc(n->src[2]);
                cllm_tensor dm; if (n->src[3]) dm = desc(n->src[3]);
                const size_t need = cllm_flash_attn_wsize(&da);
                if ((rc = ensure_wdata(c, need))) break;
                rc = CALL(cllm_op_flash_attn_ext, st, &da, &db, &dc, n->src[3] ? &dm : nullptr, &d, scale, max_bias, softcap, c->wdata, c->wsize);
            } break;
            case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: if (plan.attn[i] >= 0) {
                const fused_attn & A = plan.attns[plan.attn[i]];
                if (A.level == 2) {
                    const bool table = A.hd == 64 || A.hd == 128;       // head sizes of the compact kernels
                    bool have = tab_hd == A.hd && tab_fb == A.freq_base && tab_pos != nullptr;
                    if (have && tab_pos != A.pos) {                     // another layer's position tensor: the same value, as far as the host's writes tell?
                        std::lock_guard<std::mutex> lock(g_ring.m);
                        const auto x = g_i32_vals.find(tab_pos), y = g_i32_vals.find(A.pos);
                        have = x != g_i32_vals.end() && y != g_i32_vals.end() && x->second == y->second;
                    }
                    if (table && !have) {
                        if ((rc = CALL(cllm_op_rope_table, st, A.pos, A.hd, A.freq_base, a_cs))) break;
                        tab_pos = A.pos; tab_hd = A.hd; tab_fb = A.freq_base;
                    }
                    rc = CALL(cllm_op_rope_kv_attn_decode, st, a_qkv, A.pos, table ? a_cs : nullptr, A.freq_base, sw ? (int64_t)(cllm_attn_decode_wsize(A.n_kv, A.nh, A.ML) != 0) : A.n_kv, A.nh, A.nkv, A.hd, A.mode, A.k_cache, A.v_cache, A.ML,
                                                     A.out, score_bytes ? a_scores : nullptr, score_bytes);
                } else if (A.alias) {
                    rc = CALL(cllm_op_attn_decode, st, A.q, A.pos, A.nh, A.nkv, A.hd, A.k_cache, A.v_cache, A.ML, a_stage);
                    if (rc == CLLM_OK) rc = CALL(cllm_memcpy_d2d, (void *) A.out, (const void *) a_stage, (size_t) A.nh * A.hd * 4, st);
                } else rc = CALL(cllm_op_attn_decode, st, A.q, A.pos, A.nh, A.nkv, A.hd, A.k_cache, A.v_cache, A.ML, A.out);
            } else rc = CALL(cllm_op_cpy, st, &da, &d); break;
            case GGML_OP_GET_ROWS: rc = CALL(cllm_op_get_rows, st, &da, &db, &d); break;
            default: HIPB_LOG("graph_compute: op %s reached the device although supports_op() declined it", ggml_op_name(n->op)); return GGML_STATUS_FAILED;
        }
        if (rc != CLLM_OK) {
            HIPB_LOG("node %d (%s, '%s') failed: %s", i, ggml_op_name(n->op), n->name, cllm_last_error());
            return rc == CLLM_E_ALLOC ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;
#undef CALL
    };
    static const bool no_graph = getenv("CLLM_HIP_GRAPH") && atoi(getenv("CLLM_HIP_GRAPH")) == 0;      // default: replay; CLLM_HIP_GRAPH=0 issues every call
    bool replayed = false;
    if (no_graph || c->graph_broken) {
        const ggml_status rs = walk(plan, nullptr);
        if (rs != GGML_STATUS_SUCCESS) return rs;
    } else {
        static thread_local std::vector<uint8_t> sig;
        sig.clear();
        fuse_plan probe = plan;                    // the walk changes the plan (merge decisions): sign on a copy
        sig_writer sw{ sig };
        static const bool sig_dbg = getenv("CLLM_HIP_SIG_DEBUG") != nullptr;
        static thread_local std::vector<size_t> marks;
        if (sig_dbg) { marks.clear(); sw.marks = &marks; }
        ggml_status rs = walk(probe, &sw);
        if (sig_dbg && sig != c->last_sig && sig.size() == c->last_sig.size()) {           // same shape of list, different bytes: which call, which byte?
            size_t off = 0; while (off < sig.size() && sig[off] == c->last_sig[off]) off++;
            size_t k = 0; while (k + 1 < marks.size() && marks[k + 1] <= off) k++;
            HIPB_LOG("launch list differs from the previous graph's in call %zu of %zu, byte %zu of that call", k, marks.size(), off - marks[k]);
        } else if (sig_dbg && sig != c->last_sig) HIPB_LOG("launch list differs from the previous graph's in length: %zu vs %zu bytes", sig.size(), c->last_sig.size());
        if (rs != GGML_STATUS_SUCCESS) return rs;
        const int n_calls = launches;
        bool ahead_hit = false;
        if (c->ahead.inflight) {                   // a step is running ahead: is it the one the host is asking for?
            const int hr = ahead_resolve(c, c->graph_exec && sig == c->graph_sig, cur_sets);
            if (hr < 0) return GGML_STATUS_FAILED;
            ahead_hit = hr == 1;
        }
        if (ahead_hit) {
            replayed = true; c->replays++;         // nothing to launch
            for (size_t k = 0; k < plan.groups.size(); k++) plan.groups[k].state = probe.groups[k].state;
        } else if (c->graph_exec && sig == c->graph_sig) {
            if (cllm_graph_launch(c->graph_exec, st) != CLLM_OK) { HIPB_LOG("graph replay failed: %s", cllm_last_error()); return GGML_STATUS_FAILED; }
            replayed = true; c->replays++;
            for (size_t k = 0; k < plan.groups.size(); k++) plan.groups[k].state = probe.groups[k].state;      // (statistics)
        } else if (n_calls >= 8 && sig == c->last_sig && cllm_graph_capture_begin(st) == CLLM_OK) {           // second time in a row: capture, then launch
            rs = walk(plan, nullptr);
            void * exec = nullptr;
            const int rc = cllm_graph_capture_end(st, &exec);
            if (rs != GGML_STATUS_SUCCESS) { if (exec) cllm_graph_destroy(exec); return rs; }
            if (rc != CLLM_OK || !exec) {           // nothing ran: do it the plain way, and stop trying
                HIPB_LOG("launch-list capture failed (%s): issuing the calls of every graph from now on", cllm_last_error());
                c->graph_broken = true;
                fuse_plan again = make_plan(g);
                for (const fused_attn & A : again.attns) if (A.level == 2) {
                    again.mvs[A.wq].dst = a_qkv; again.mvs[A.wk].dst = a_qkv + (size_t) A.hd * A.nh; again.mvs[A.wv].dst = a_qkv + (size_t) A.hd * (A.nh + A.nkv);
                }
                rs = walk(again, nullptr);
                if (rs != GGML_STATUS_SUCCESS) return rs;
            } else {
                if (c->graph_exec) { cllm_stream_sync(st); cllm_graph_destroy(c->graph_exec); }
                c->graph_exec = exec; c->graph_sig = sig; c->captures++;
                if (cllm_graph_launch(exec, st) != CLLM_OK) { HIPB_LOG("graph launch failed: %s", cllm_last_error()); return GGML_STATUS_FAILED; }
            }
        } else {
            rs = walk(plan, nullptr);
            if (rs != GGML_STATUS_SUCCESS) return rs;
        }
        c->last_sig.swap(sig);
    }
    ahead_arm(c, g, plan, cur_sets, replayed && c->graph_exec != nullptr, false);
    if (g_stats) {
        int a1 = 0, a2 = 0;
        for (const fused_attn & A : plan.attns) (A.level == 2 ? a2 : a1)++;
        for (const merge_group & G : plan.groups) if (G.state == 1) launches -= G.n - 1;
        int merged_n = 0;
        for (const merge_group & G : plan.groups) merged_n += G.state == 1;
        int fa_nodes = 0;
        for (int i = 0; i < ggml_graph_n_nodes(g); i++) fa_nodes += ggml_graph_node(g, i)->op == GGML_OP_FLASH_ATTN_EXT;
        HIPB_LOG("%s graph_compute: %d nodes -> %d calls%s (%d fused mat-vecs, %d merged over packed weights, attention fused at level 1: %d, level 2: %d, flash prefill: %d, flash_attn_ext: %d, MoE routers: %d, prefill mat-muls with fused prologue / epilogue: %d)",
                 be_name(backend), ggml_graph_n_nodes(g), launches, replayed ? ", replayed from the captured graph" : "", (int) plan.mvs.size(), merged_n, a1, a2, (int) plan.fas.size(), fa_nodes,
                 (int) plan.routers.size(), (int) plan.pfs.size());
        g_ws.calls += launches;
        if (++g_ws.graphs % 64 == 0) {
            const double n = 64.0;
            HIPB_LOG("per graph over the last 64: host %.0f us | graph_compute %.0f us (of which planning %.0f us, %.0f calls) | synchronize %.0f us (%.1f calls) | "
                     "set_tensor %.0f us (%.1f calls) | get_tensor %.0f us (%.1f calls) | buffer alloc/free %.0f us (%.2f calls)",
                     g_ws.host_us / n, g_ws.issue_us / n, g_ws.plan_us / n, g_ws.calls / n, g_ws.sync_us / n, g_ws.syncs / n, g_ws.set_us / n, g_ws.sets / n, g_ws.get_us / n, g_ws.gets / n,
                     g_ws.alloc_us / n, g_ws.allocs / n);
            HIPB_LOG("launch lists replayed from a captured graph so far: %ld (captures: %ld); steps started ahead of the host: %ld, of which the host then asked for: %ld", c->replays, c->captures, c->ahead.launched, c->ahead.hits);
            g_ws.host_us = g_ws.plan_us = g_ws.issue_us = g_ws.sync_us = g_ws.set_us = g_ws.get_us = g_ws.alloc_us = 0; g_ws.calls = g_ws.sets = g_ws.gets = g_ws.allocs = g_ws.syncs = 0;
        }
    }
    return GGML_STATUS_SUCCESS;       // asynchronous: the host calls synchronize() before it reads (src/backend.cpp:824-825)
}

// ---- events and the asynchronous copy between two backends of this module (ggml-backend-impl.h:87-127).  The scheduler uses them for the
//      activations that cross a layer split (`-ngl 0:40;1:40`, ggml-backend.cpp:414-433, 1473-1477): without them it synchronizes both devices and
//      goes through the blocking buffer copy at every boundary. ----
bool be_is_ours(ggml_backend_t b);
void be_event_record(ggml_backend_t b, ggml_backend_event_t e) {
    auto * c = (hip_backend_ctx *) b->context; cllm_set_device(c->device);
    if (cllm_event_record(e->context, c->stream) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] event_record failed: %s\n", cllm_last_error());
}
void be_event_wait(ggml_backend_t b, ggml_backend_event_t e) {
    auto * c = (hip_backend_ctx *) b->context; cllm_set_device(c->device);
    if (cllm_stream_wait_event(c->stream, e->context) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] event_wait failed: %s\n", cllm_last_error());
}
// src is complete once everything queued on backend_src so far has run; backend_dst must not touch dst before the copy has landed:
// the copy goes on the SOURCE stream, an event recorded behind it is what the destination stream waits for.
bool be_cpy_tensor_async(ggml_backend_t bs, ggml_backend_t bd, const ggml_tensor * src, ggml_tensor * dst) {
    if (!be_is_ours(bs) || !be_is_ours(bd) || !src->buffer || !dst->buffer || src->buffer->iface.get_base != buf_base || dst->buffer->iface.get_base != buf_base) return false;
    if (!ggml_is_contiguous(src) || !ggml_is_contiguous(dst) || ggml_nbytes(src) != ggml_nbytes(dst)) return false;
    auto * cs = (hip_backend_ctx *) bs->context; auto * cd = (hip_backend_ctx *) bd->context;
    auto * xs = (hip_buffer_ctx *) src->buffer->context; auto * xd = (hip_buffer_ctx *) dst->buffer->context;
    if (xs->device != cs->device || xd->device != cd->device) return false;       // (a tensor living on a third device: the blocking path)
    flush_sets();                                  // staged host writes to either tensor are ordered before the copy
    xd->gen++; g_tp_kv_epoch++;
    { std::lock_guard<std::mutex> lock(g_ring.m); i32_forget(dst->data, ggml_nbytes(dst)); }
    cllm_set_device(cs->device);
    if (cllm_memcpy_peer_async(dst->data, cd->device, src->data, cs->device, ggml_nbytes(src), cs->stream) != CLLM_OK) { GGML_LOG_ERROR("[ggml-hip] cpy_tensor_async ('%s' -> '%s') failed: %s\n", src->name, dst->name, cllm_last_error()); return false; }
    if (bs != bd) {
        if (!cs->copy_event) { if (cllm_event_create(&cs->copy_event) != CLLM_OK) { cs->copy_event = nullptr; cllm_stream_sync(cs->stream); return true; } }
        if (cllm_event_record(cs->copy_event, cs->stream) != CLLM_OK) { cllm_stream_sync(cs->stream); return true; }
        cllm_set_device(cd->device);
        if (cllm_stream_wait_event(cd->stream, cs->copy_event) != CLLM_OK) { cllm_set_device(cs->device); cllm_stream_sync(cs->stream); }
    }
    return true;
}
const ggml_backend_i k_backend_i = {
    be_name, be_free,
    nullptr, nullptr,                   // set/get_tensor_async: the buffer calls stage through the pinned ring already
    be_cpy_tensor_async,
    be_sync,
    nullptr, nullptr, nullptr, nullptr, // graph plans
    be_graph_compute,
    be_event_record, be_event_wait,
    nullptr,                            // graph_optimize
};
bool be_is_ours(ggml_backend_t b) { return b && b->iface.get_name == be_name; }
ggml_guid k_guid = { 0x63, 0x6c, 0x6c, 0x6d, 0x2d, 0x68, 0x69, 0x70, 0x2d, 0x67, 0x66, 0x78, 0x39, 0x35, 0x30, 0x01 };

// ---------------------------------------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------------------------------------
const char * dev_name(ggml_backend_dev_t d) { return ((hip_device_ctx *) d->context)->name.c_str(); }
const char * dev_desc(ggml_backend_dev_t d) { return ((hip_device_ctx *) d->context)->desc.c_str(); }
void dev_memory(ggml_backend_dev_t d, size_t * free, size_t * total) { cllm_device_info(((hip_device_ctx *) d->context)->id, nullptr, 0, free, total, nullptr); }
enum ggml_backend_dev_type dev_type(ggml_backend_dev_t) { return GGML_BACKEND_DEVICE_TYPE_GPU; }
void dev_props(ggml_backend_dev_t d, ggml_backend_dev_props * p) {
    p->name = dev_name(d); p->description = dev_desc(d); p->type = GGML_BACKEND_DEVICE_TYPE_GPU; p->device_id = nullptr;
    dev_memory(d, &p->memory_free, &p->memory_total);
    p->caps = { /*async*/ true, /*host_buffer*/ false, /*buffer_from_host_ptr*/ false, /*events*/ true };
}
ggml_backend_t dev_init(ggml_backend_dev_t d, const char *) {
    auto * dc = (hip_device_ctx *) d->context;
    if (cllm_set_device(dc->id)) { HIPB_LOG("set_device failed: %s", cllm_last_error()); return nullptr; }
    auto * c = new hip_backend_ctx{ dc->id };
    if (cllm_stream_create(&c->stream)) { HIPB_LOG("stream_create failed: %s", cllm_last_error()); delete c; return nullptr; }
    return new ggml_backend{ &k_guid, k_backend_i, d, c };
}
ggml_backend_buffer_type_t dev_buft(ggml_backend_dev_t d) { return &((hip_device_ctx *) d->context)->buft; }
bool dev_supports_buft(ggml_backend_dev_t d, ggml_backend_buffer_type_t t) { return t->iface.get_name == buft_name && t->context == d->context; }
ggml_backend_event_t dev_event_new(ggml_backend_dev_t d) {
    cllm_set_device(((hip_device_ctx *) d->context)->id);
    void * e = nullptr;
    if (cllm_event_create(&e) != CLLM_OK) { GGML_LOG_ERROR("[ggml-hip] event_new failed: %s\n", cllm_last_error()); return nullptr; }
    return new ggml_backend_event{ d, e };
}
void dev_event_free(ggml_backend_dev_t, ggml_backend_event_t e) { if (e) { cllm_event_destroy(e->context); delete e; } }
void dev_event_synchronize(ggml_backend_dev_t, ggml_backend_event_t e) { if (cllm_event_sync(e->context) != CLLM_OK) GGML_LOG_ERROR("[ggml-hip] event_synchronize failed: %s\n", cllm_last_error()); }
const ggml_backend_device_i k_device_i = {
    dev_name, dev_desc, dev_memory, dev_type, dev_props, dev_init, dev_buft,
    nullptr, nullptr,                   // host buffer type, buffer_from_host_ptr (SURVEY.md 8b note: keep NULL)
    dev_supports_op, dev_supports_buft,
    nullptr,                            // offload_op
    dev_event_new, dev_event_free, dev_event_synchronize,
};
std::vector<ggml_backend_device> g_dev_objs;

// ---------------------------------------------------------------------------------------------------------------------------
// reg
// ---------------------------------------------------------------------------------------------------------------------------
const char * reg_name(ggml_backend_reg_t) { return "HIP"; }
size_t reg_count(ggml_backend_reg_t) { return g_dev_objs.size(); }
ggml_backend_dev_t reg_get(ggml_backend_reg_t, size_t i) { return i < g_dev_objs.size() ? &g_dev_objs[i] : nullptr; }
void * reg_proc(ggml_backend_reg_t, const char *) { return nullptr; }
const ggml_backend_reg_i k_reg_i = { reg_name, reg_count, reg_get, reg_proc };

}  // namespace

extern "C" {
GGML_BACKEND_API ggml_backend_reg_t ggml_backend_init(void);
GGML_BACKEND_API int ggml_backend_score(void);
}

ggml_backend_reg_t ggml_backend_init(void) {
    static std::once_flag once;
    std::call_once(once, [] {
        // CLLM_HIP_VIRTUAL_DEVICES=n: register n ggml devices over the physical ones (device i runs on GPU i % physical) -- the reference's own multi-device modes
        // (layer split `-ngl "0:16;1:16"`, src/backend.cpp:677-778: a buffer type, a backend, a stream per device; activations cross through cpy_tensor_async +
        // events) then run through the real scheduler on a one-GPU box.  Each ggml device has its own buffer type, so tensors of device 1 are foreign to device 0
        // exactly as on two GPUs; what the physical sharing changes is only that the "peer" copy stays inside one HBM.
        const int phys = cllm_device_count();
        int n = phys;
        if (const char * v = getenv("CLLM_HIP_VIRTUAL_DEVICES")) { const int k = atoi(v); if (k > 0 && phys > 0) n = k > 64 ? 64 : k; }
        // CLLM_HIP_TP=N: ONE logical device over N tensor-parallel ranks (rank r on GPU r % physical: more ranks than GPUs = virtual ranks sharing a GPU) -- tp_graph_compute
        if (const char * v = getenv("CLLM_HIP_TP")) { const int k = atoi(v); if (k > 1 && phys > 0) { g_tp.n = k > 16 ? 16 : k; n = 1; } }
        g_dev_objs.reserve(n);
        for (int i = 0; i < n; i++) {
            auto * d = new hip_device_ctx();
            char name[256] = "MI355X"; int cus = 0;
            const int gpu = i % phys;
            cllm_device_info(gpu, name, sizeof(name), nullptr, nullptr, &cus);
            d->id = gpu; d->name = "HIP" + std::to_string(i); d->desc = std::string(name) + ", " + std::to_string(cus) + " CUs (chatllm.cpp_amd" + (n != phys ? ", GPU " + std::to_string(gpu) : std::string()) +
                      (g_tp.n > 1 ? ", tensor parallel over " + std::to_string(g_tp.n) + " ranks on " + std::to_string(g_tp.n < phys ? g_tp.n : phys) + " GPU(s)" : std::string()) + ")";
            g_devices.push_back(d);
            g_dev_objs.push_back(ggml_backend_device{ k_device_i, &g_reg, d });
            d->buft = ggml_backend_buffer_type{ k_buft_i, &g_dev_objs.back(), d };
        }
        g_reg = ggml_backend_reg{ GGML_BACKEND_API_VERSION, k_reg_i, nullptr };
    });
    return &g_reg;
}
int ggml_backend_score(void) { return cllm_device_count() > 0 ? 100 : 0; }
