// decoder.hip -- host-side runner for the Llama-3 / Qwen2 decoder block sequence.
//
// Mirrors (node for node) what chatllm.cpp builds per graph:
//   HeterogeneousModel::forward (src/models.cpp:1399-1424) -> Embedding::forward (src/layers.cpp:2038-2055)
//   -> n_layer x LMBlock1Forward::forward (src/layers.cpp:2719-2761) [RMSNorm :2216, BaseAttention::forward :3212,
//      cross_attention :2681, save_to_cache :3044, calc_attn_scores :2541, BaseMLP::forward :2475]
//   -> LMFinalSteps::forward (src/models.cpp:1736-1784)
// but as a fixed launch sequence over the cllm_op_* C ABI on one HIP stream instead of a ggml graph that is
// rebuilt and re-planned for every token.  All buffers are allocated once.
//
// Two paths share the same arithmetic:
//   general (qlen >= 1): one op per graph node, exactly the nodes of SURVEY.md 3.3
//   decode  (qlen == 1): fused launches (norm+quant, rope+kv-write, attention, silu*up+quant, GEMV+residual)
//                        whose only per-token inputs (token id, position) live in device memory, so the whole
//                        step is captured once in a hipGraph and replayed.
#include "common.h"

#include <string>
#include <vector>
#include <map>
#include <math.h>
#include <string.h>
#include <stdio.h>

struct dweight { int type = -1; void * data = nullptr; size_t bytes = 0; bool owned = false; };

struct llama_layer {
    dweight attn_norm, ffn_norm, wq, wk, wv, wo, wgate, wup, wdown, bq, bk, bv;
    dweight wqkv, wgu;            // row fusions, numerically identical, one launch: q|k|v concatenated; gate/up INTERLEAVED (row 2u = gate_u,
                                  // row 2u+1 = up_u) so that one wave owns both rows of a feature and can apply SiLU(gate)*up itself
    dweight bqkv;                 // concatenated q|k|v bias (Qwen2)
    uint16_t * k_cache = nullptr, * v_cache = nullptr;
};

struct cllm_llama {
    cllm_llama_config cfg;
    hipStream_t st = nullptr;
    int nh = 0, nkv = 0, F = 0;                  // local (tensor-parallel) sizes
    dweight tok_embd, lm_head, out_norm;
    std::vector<llama_layer> layers;
    bool finalized = false;
    // activations
    int maxq = 0;
    float * x = nullptr, * xn = nullptr, * qkv = nullptr, * att = nullptr, * ctx = nullptr, * o = nullptr, * gu = nullptr, * g = nullptr;
    float * scores = nullptr; size_t scores_elems = 0;
    float * logits = nullptr;
    void *  wdata = nullptr; size_t wsize = 0;
    int32_t * tokens_dev = nullptr, * pos_dev = nullptr;
    cllm_allreduce_fn allreduce = nullptr; void * allreduce_user = nullptr;
    void * tp_comm = nullptr;         // RCCL communicator (cllm_tp_init): the all-reduce runs on the runner's stream, inside the decode graph
    void * tp_oneshot = nullptr;      // one-shot direct-write all-reduce over IPC-mapped peer buffers (tp_oneshot.hip): one launch, also inside the graph
    void * tp_fused = nullptr;        // the decode steps' all-reduces FUSED into the neighbouring mat-vecs (gemv_tp.hip): no all-reduce launch; prompts keep the collective above
    bool use_graph = true;
    hipGraphExec_t decode_graph = nullptr;       // one sampled decode step, short-context attention (one launch per layer)
    hipGraphExec_t decode_graph_long = nullptr;  // the same with the split long-context attention (attn_long.hip)
    int32_t * next_tok_dev = nullptr;            // greedy feedback
    int32_t * out_ring = nullptr, * counter_dev = nullptr;   // device-side greedy loop: generated ids + how many
    bool own_stream = false, fused_ok = false, fused_warm = false, fused_warm_long = false;
    size_t weight_bytes = 0;
    void * ffn_state = nullptr; bool ffn_state_ready = false;          // ffn_fused.hip: epoch word + granules of the fused FFN launch
};

#define TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

static cllm_tensor T(int type, void * data, int64_t n0, int64_t n1 = 1, int64_t n2 = 1, int64_t n3 = 1) {
    cllm_tensor t; t.type = type; t.data = data;
    t.ne[0] = n0; t.ne[1] = n1; t.ne[2] = n2; t.ne[3] = n3;
    t.nb[0] = cllm_type_size(type); t.nb[1] = cllm_row_size(type, n0); t.nb[2] = t.nb[1] * (size_t) n1; t.nb[3] = t.nb[2] * (size_t) n2;
    return t;
}
static cllm_tensor TS(int type, void * data, int64_t n0, int64_t n1, int64_t n2, size_t nb1, size_t nb2) {
    cllm_tensor t = T(type, data, n0, n1, n2);
    t.nb[1] = nb1; t.nb[2] = nb2; t.nb[3] = nb2 * (size_t) n2;
    return t;
}

extern "C" int cllm_tp_split(int64_t n_units, int tp_size, int tp_rank, int64_t * first, int64_t * count) {
    if (n_units < 0 || tp_size < 1 || tp_rank < 0 || tp_rank >= tp_size || !first || !count) FAIL(CLLM_E_INVALID, "tp_split: arguments");
    const int64_t base = n_units / tp_size, extra = n_units % tp_size;
    *count = base + (tp_rank < extra ? 1 : 0);
    *first = base * tp_rank + (tp_rank < extra ? tp_rank : extra);
    return CLLM_OK;
}

extern "C" int cllm_llama_create(const cllm_llama_config * cfg, void * stream, cllm_llama ** out) {
    if (!cfg || !out) FAIL(CLLM_E_INVALID, "llama_create: null");
    if (cfg->n_layer <= 0 || cfg->hidden <= 0 || cfg->n_head <= 0 || cfg->n_kv_head <= 0 || cfg->head_dim <= 0 || cfg->ffn <= 0 || cfg->vocab <= 0 || cfg->max_len <= 0)
        FAIL(CLLM_E_INVALID, "llama_create: bad config");
    const int tp = cfg->tp_size > 0 ? cfg->tp_size : 1;
    if (cfg->n_head % tp || cfg->n_kv_head % tp || cfg->n_head % cfg->n_kv_head) FAIL(CLLM_E_INVALID, "llama_create: heads not divisible by tp_size");
    if (cfg->ffn_local < 0 || cfg->ffn_local > cfg->ffn) FAIL(CLLM_E_INVALID, "llama_create: ffn_local %d outside 0..ffn", cfg->ffn_local);
    if (cfg->ffn_local == 0 && cfg->ffn % tp) FAIL(CLLM_E_INVALID, "llama_create: ffn %d is not divisible by tp_size %d: pass this rank's share as ffn_local (cllm_tp_split over the down projection's quant blocks)", cfg->ffn, tp);
    cllm_llama * m = new cllm_llama();
    m->cfg = *cfg; m->cfg.tp_size = tp;
    m->st = (hipStream_t) stream;
    if (!m->st) {   // graphs cannot be captured on the legacy NULL stream: own a blocking stream (it still orders against NULL-stream work)
        if (hipStreamCreate(&m->st) != hipSuccess) { (void) hipGetLastError(); m->st = nullptr; } else m->own_stream = true;
    }
    m->nh = cfg->n_head / tp; m->nkv = cfg->n_kv_head / tp; m->F = cfg->ffn_local > 0 ? cfg->ffn_local : cfg->ffn / tp;
    m->layers.resize(cfg->n_layer);
    *out = m;
    return CLLM_OK;
}

static void free_w(dweight & w) { if (w.owned && w.data) (void) hipFree(w.data); w = dweight(); }

extern "C" void cllm_llama_destroy(cllm_llama * m) {
    if (!m) return;
    (void) hipStreamSynchronize(m->st);
    if (m->decode_graph) (void) hipGraphExecDestroy(m->decode_graph);
    if (m->decode_graph_long) (void) hipGraphExecDestroy(m->decode_graph_long);
    free_w(m->tok_embd); free_w(m->lm_head); free_w(m->out_norm);
    for (auto & L : m->layers) {
        for (dweight * w : { &L.attn_norm, &L.ffn_norm, &L.wq, &L.wk, &L.wv, &L.wo, &L.wgate, &L.wup, &L.wdown, &L.bq, &L.bk, &L.bv, &L.wqkv, &L.wgu, &L.bqkv }) free_w(*w);
        if (L.k_cache) (void) hipFree(L.k_cache);
        if (L.v_cache) (void) hipFree(L.v_cache);
    }
    for (void * p : { (void *) m->x, (void *) m->xn, (void *) m->qkv, (void *) m->att, (void *) m->ctx, (void *) m->o, (void *) m->gu, (void *) m->g, (void *) m->scores,
                      (void *) m->logits, m->wdata, (void *) m->tokens_dev, (void *) m->pos_dev, (void *) m->next_tok_dev, (void *) m->out_ring, (void *) m->counter_dev, m->ffn_state }) if (p) (void) hipFree(p);
    if (m->own_stream) { stream_scratch_release(m->st); (void) hipStreamDestroy(m->st); }
    delete m;
}

static dweight * find_weight(cllm_llama * m, const char * name) {
    std::string n(name);
    if (n == "tok_embd") return &m->tok_embd;
    if (n == "lm_head")  return &m->lm_head;
    if (n == "out_norm") return &m->out_norm;
    if (n.rfind("layers.", 0) == 0) {
        const size_t dot = n.find('.', 7);
        if (dot == std::string::npos) return nullptr;
        const int il = atoi(n.substr(7, dot - 7).c_str());
        if (il < 0 || il >= (int) m->layers.size()) return nullptr;
        llama_layer & L = m->layers[il];
        const std::string f = n.substr(dot + 1);
        static const std::map<std::string, dweight llama_layer::*> fields = {
            {"attn_norm", &llama_layer::attn_norm}, {"ffn_norm", &llama_layer::ffn_norm}, {"wq", &llama_layer::wq}, {"wk", &llama_layer::wk},
            {"wv", &llama_layer::wv}, {"wo", &llama_layer::wo}, {"wgate", &llama_layer::wgate}, {"wup", &llama_layer::wup}, {"wdown", &llama_layer::wdown},
            {"bq", &llama_layer::bq}, {"bk", &llama_layer::bk}, {"bv", &llama_layer::bv}, {"wqkv", &llama_layer::wqkv}, {"wgu", &llama_layer::wgu},
        };
        auto it = fields.find(f);
        return it == fields.end() ? nullptr : &(L.*(it->second));
    }
    return nullptr;
}

static int set_w(cllm_llama * m, const char * name, int type, void * data, size_t nbytes, bool copy_from_host) {
    if (!m || !name || !data) FAIL(CLLM_E_INVALID, "llama_set_weight: null");
    if (m->finalized) FAIL(CLLM_E_INVALID, "llama_set_weight: model already running");
    dweight * w = find_weight(m, name);
    if (!w) FAIL(CLLM_E_INVALID, "llama_set_weight: unknown tensor '%s'", name);
    if (cllm_type_size(type) == 0) FAIL(CLLM_E_UNSUPPORTED, "llama_set_weight: type %d", type);
    free_w(*w);
    w->type = type; w->bytes = nbytes;
    if (copy_from_host) {
        HIP_TRY(hipMalloc(&w->data, nbytes));
        w->owned = true;
        HIP_TRY(hipMemcpy(w->data, data, nbytes, hipMemcpyHostToDevice));
    } else { w->data = data; w->owned = false; }
    return CLLM_OK;
}
extern "C" int cllm_llama_set_weight(cllm_llama * m, const char * name, int type, const void * data, size_t nbytes) { return set_w(m, name, type, (void *) data, nbytes, true); }
extern "C" int cllm_llama_bind_weight(cllm_llama * m, const char * name, int type, void * dev, size_t nbytes) { return set_w(m, name, type, dev, nbytes, false); }
extern "C" int cllm_llama_set_allreduce(cllm_llama * m, cllm_allreduce_fn fn, void * user) { if (!m) FAIL(CLLM_E_INVALID, "null"); m->allreduce = fn; m->allreduce_user = user; return CLLM_OK; }
extern "C" int cllm_llama_set_tp_comm(cllm_llama * m, void * comm) { if (!m) FAIL(CLLM_E_INVALID, "null"); m->tp_comm = comm; return CLLM_OK; }
extern "C" int cllm_llama_set_tp_oneshot(cllm_llama * m, void * os) { if (!m) FAIL(CLLM_E_INVALID, "null"); m->tp_oneshot = os; return CLLM_OK; }
extern "C" const void * cllm_tp_fused_dev(void * os);
extern "C" int cllm_tp_fused_sites(void * os);
extern "C" size_t cllm_tp_fused_max_n(void * os);
extern "C" int cllm_tp_fused_advance(void * os, void * stream);
extern "C" int cllm_tp_fused_error(void * os);
extern "C" int cllm_tp_fused_fine_grained(void * os);
// bind the receive buffers of the fused all-reduce (cllm_tp_fused_create / _connect): 2 n_layer sites of `hidden` values.  The single-token steps then run without
// all-reduce launches; multi-token graphs (prompts) still need cllm_llama_set_tp_comm / _set_tp_oneshot / _set_allreduce.  NULL: back to the collective per all-reduce.
extern "C" int cllm_llama_set_tp_fused(cllm_llama * m, void * os) {
    if (!m) FAIL(CLLM_E_INVALID, "null");
    if (os && (cllm_tp_fused_sites(os) < 2 * m->cfg.n_layer || cllm_tp_fused_max_n(os) < (size_t) m->cfg.hidden))
        FAIL(CLLM_E_INVALID, "llama_set_tp_fused: the buffers hold %d sites of %zu values, the model needs %d of %d", cllm_tp_fused_sites(os), cllm_tp_fused_max_n(os), 2 * m->cfg.n_layer, m->cfg.hidden);
    // ranks of DIFFERENT processes that share one GPU (coarse-grained buffers, CLLM_TP_ONESHOT_SAME_DEVICE=1: a test vehicle) poll for each other's launches from
    // separate queues -- that only terminates if both ranks' 1024-thread workgroups can be resident at once, i.e. at small shapes.  At hidden sizes whose gather launch fills the
    // GPU by itself the peer's scatter cannot start and every step would end in the bounded wait's time-out: refuse instead.  (Ranks inside one process on one GPU -- the ggml
    // module's virtual ranks -- share a stream and are issued site by site: no such limit.)
    if (os && cllm_tp_fused_fine_grained(os) == 0 && m->cfg.tp_size > 1 && m->cfg.hidden > 2048 && !getenv("CLLM_TP_FUSED_SAME_DEVICE_ANY_SIZE"))
        FAIL(CLLM_E_UNSUPPORTED, "llama_set_tp_fused: ranks of different processes on ONE GPU (coarse-grained receive buffers) cannot co-reside at hidden %d: the fused all-reduce needs distinct GPUs "
                                 "(fine-grained buffers) here; use the collective (cllm_llama_set_tp_oneshot / _set_tp_comm), or CLLM_TP_FUSED_SAME_DEVICE_ANY_SIZE=1 to try anyway", m->cfg.hidden);
    if (m->decode_graph) { (void) hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }            // (captured with the other form of the all-reduce)
    if (m->decode_graph_long) { (void) hipGraphExecDestroy(m->decode_graph_long); m->decode_graph_long = nullptr; }
    m->tp_fused = os;
    m->fused_warm = m->fused_warm_long = false;       // the next step runs eagerly once more (the new launches set their function attributes outside a capture)
    return CLLM_OK;
}
extern "C" int cllm_tp_all_reduce_f32(void * comm, void * stream, float * buf, size_t n);
extern "C" int cllm_tp_oneshot_all_reduce_f32(void * os, void * stream, float * buf, size_t n);
extern "C" int cllm_tp_oneshot_error(void * os);
// sum `n` floats of `buf` over the tensor-parallel group, stream-ordered: RCCL if a communicator is bound, else the host callback
static int tp_allreduce(cllm_llama * m, hipStream_t st, float * buf, int64_t n) {
    if (m->tp_oneshot) {                                           // decode-sized messages; larger ones (a prompt's [H, qlen]) fall through to RCCL / the callback
        const int rc = cllm_tp_oneshot_all_reduce_f32(m->tp_oneshot, st, buf, (size_t) n);
        if (rc != CLLM_E_UNSUPPORTED) return rc;
    }
    if (m->tp_comm) return cllm_tp_all_reduce_f32(m->tp_comm, st, buf, (size_t) n);
    if (m->allreduce) { m->allreduce(m->allreduce_user, st, buf, n); return CLLM_OK; }
    FAIL(CLLM_E_INVALID, "llama: tp_size > 1 needs cllm_llama_set_tp_comm or cllm_llama_set_allreduce (cllm_llama_set_tp_oneshot covers messages up to its max_n)");
}
// a sharded model: o / down produce PARTIAL sums.  (tp_fused alone carries the single-token steps only: a prompt through forward_general still needs a collective --
// tp_allreduce fails loudly without one instead of adding partial sums to the residual stream)
static bool tp_on(const cllm_llama * m) { return m->cfg.tp_size > 1 && (m->tp_comm || m->tp_oneshot || m->allreduce || m->tp_fused); }
extern "C" int cllm_llama_use_graph(cllm_llama * m, int enable) { if (!m) FAIL(CLLM_E_INVALID, "null"); m->use_graph = enable != 0; return CLLM_OK; }
extern "C" size_t cllm_llama_weight_bytes(const cllm_llama * m) { return m ? m->weight_bytes : 0; }

static int expect(const dweight & w, const char * what, int il, size_t bytes, bool f32only) {
    if (!w.data) FAIL(CLLM_E_INVALID, "llama: missing tensor %s (layer %d)", what, il);
    if (f32only && w.type != CLLM_TYPE_F32) FAIL(CLLM_E_INVALID, "llama: %s must be F32", what);
    if (bytes && w.bytes != bytes) FAIL(CLLM_E_INVALID, "llama: %s (layer %d) has %zu bytes, expected %zu", what, il, w.bytes, bytes);
    return CLLM_OK;
}

// concatenate rows of a and b (and c) into one device buffer
static int fuse_rows(hipStream_t st, dweight & dst, std::initializer_list<dweight *> parts) {
    size_t total = 0; int type = -1;
    for (dweight * p : parts) { if (type < 0) type = p->type; if (p->type != type) return CLLM_OK; total += p->bytes; }   // mixed types: keep separate
    void * buf = nullptr;
    HIP_TRY(hipMalloc(&buf, total));
    size_t off = 0;
    for (dweight * p : parts) { HIP_TRY(hipMemcpyAsync((char *) buf + off, p->data, p->bytes, hipMemcpyDeviceToDevice, st)); off += p->bytes; }
    HIP_TRY(hipStreamSynchronize(st));
    for (dweight * p : parts) free_w(*p);
    dst.type = type; dst.data = buf; dst.bytes = total; dst.owned = true;
    return CLLM_OK;
}

// dst rows alternate a_0, b_0, a_1, b_1, ... (a and b have the same type and row size)
static int interleave_rows(hipStream_t st, dweight & dst, dweight & a, dweight & b, size_t row_bytes) {
    if (a.type != b.type || a.bytes != b.bytes || a.bytes % row_bytes) return CLLM_OK;     // mixed: keep separate
    void * buf = nullptr;
    HIP_TRY(hipMalloc(&buf, a.bytes + b.bytes));
    const size_t rows = a.bytes / row_bytes;
    HIP_TRY(hipMemcpy2DAsync(buf, 2 * row_bytes, a.data, row_bytes, row_bytes, rows, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemcpy2DAsync((char *) buf + row_bytes, 2 * row_bytes, b.data, row_bytes, row_bytes, rows, hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipStreamSynchronize(st));
    dst.type = a.type; dst.data = buf; dst.bytes = a.bytes + b.bytes; dst.owned = true;
    free_w(a); free_w(b);
    return CLLM_OK;
}

static int finalize(cllm_llama * m, int qlen) {
    const cllm_llama_config & c = m->cfg;
    // a sharded model without a collective would silently produce logits from partial o / down sums
    if (c.tp_size > 1 && !m->tp_comm && !m->tp_oneshot && !m->allreduce && !(m->tp_fused && qlen == 1)) FAIL(CLLM_E_INVALID, "llama: tp_size %d needs cllm_llama_set_tp_comm, cllm_llama_set_tp_oneshot or cllm_llama_set_allreduce before the first forward (cllm_llama_set_tp_fused alone carries single-token steps only)", c.tp_size);
    const int64_t H = c.hidden, hd = c.head_dim, QD = (int64_t) m->nh * hd, KD = (int64_t) m->nkv * hd, F = m->F, V = c.vocab, ML = c.max_len;
    if (!m->finalized) {
        TRY(expect(m->tok_embd, "tok_embd", -1, cllm_row_size(m->tok_embd.type, H) * (size_t) V, false));
        TRY(expect(m->lm_head, "lm_head", -1, cllm_row_size(m->lm_head.type, H) * (size_t) V, false));
        TRY(expect(m->out_norm, "out_norm", -1, (size_t) H * 4, true));
        m->weight_bytes = m->lm_head.bytes + m->out_norm.bytes;     // bytes touched per decoded token (embedding: one row)
        for (int il = 0; il < c.n_layer; il++) {
            llama_layer & L = m->layers[il];
            TRY(expect(L.attn_norm, "attn_norm", il, (size_t) H * 4, true));
            TRY(expect(L.ffn_norm, "ffn_norm", il, (size_t) H * 4, true));
            if (!L.wqkv.data) {
                TRY(expect(L.wq, "wq", il, cllm_row_size(L.wq.type, H) * (size_t) QD, false));
                TRY(expect(L.wk, "wk", il, cllm_row_size(L.wk.type, H) * (size_t) KD, false));
                TRY(expect(L.wv, "wv", il, cllm_row_size(L.wv.type, H) * (size_t) KD, false));
                TRY(fuse_rows(m->st, L.wqkv, { &L.wq, &L.wk, &L.wv }));
            } else TRY(expect(L.wqkv, "wqkv", il, cllm_row_size(L.wqkv.type, H) * (size_t)(QD + 2*KD), false));
            TRY(expect(L.wo, "wo", il, cllm_row_size(L.wo.type, QD) * (size_t) H, false));
            if (!L.wgu.data) {
                TRY(expect(L.wgate, "wgate", il, cllm_row_size(L.wgate.type, H) * (size_t) F, false));
                TRY(expect(L.wup, "wup", il, cllm_row_size(L.wup.type, H) * (size_t) F, false));
                TRY(interleave_rows(m->st, L.wgu, L.wgate, L.wup, cllm_row_size(L.wgate.type, H)));
            } else TRY(expect(L.wgu, "wgu", il, cllm_row_size(L.wgu.type, H) * (size_t)(2*F), false));
            // a rank's share of the FFN is cut in WHOLE quant blocks of the down projection's type (cllm_row_size truncates F / block: a misaligned share would pass
            // the byte check below with a matching truncated buffer and run the kernels with K % block != 0), and in groups of 8 features for the SiLU epilogue
            if (c.tp_size > 1 && (F % cllm_blck_size(L.wdown.type) || F % 8))
                FAIL(CLLM_E_INVALID, "llama: layer %d: this rank's ffn share %lld is not a whole number of the down projection's quant blocks (%d) and of 8 features; cut with cllm_tp_split",
                     il, (long long) F, cllm_blck_size(L.wdown.type));
            TRY(expect(L.wdown, "wdown", il, cllm_row_size(L.wdown.type, F) * (size_t) H, false));
            if (c.qkv_bias) {
                TRY(expect(L.bq, "bq", il, (size_t) QD * 4, true)); TRY(expect(L.bk, "bk", il, (size_t) KD * 4, true)); TRY(expect(L.bv, "bv", il, (size_t) KD * 4, true));
                void * bb = nullptr;
                HIP_TRY(hipMalloc(&bb, (size_t)(QD + 2*KD) * 4));
                HIP_TRY(hipMemcpyAsync(bb, L.bq.data, (size_t) QD * 4, hipMemcpyDeviceToDevice, m->st));
                HIP_TRY(hipMemcpyAsync((char *) bb + QD * 4, L.bk.data, (size_t) KD * 4, hipMemcpyDeviceToDevice, m->st));
                HIP_TRY(hipMemcpyAsync((char *) bb + (QD + KD) * 4, L.bv.data, (size_t) KD * 4, hipMemcpyDeviceToDevice, m->st));
                L.bqkv.type = CLLM_TYPE_F32; L.bqkv.data = bb; L.bqkv.bytes = (size_t)(QD + 2*KD) * 4; L.bqkv.owned = true;
            }
            for (const dweight * w : { &L.attn_norm, &L.ffn_norm, &L.wq, &L.wk, &L.wv, &L.wqkv, &L.wo, &L.wgate, &L.wup, &L.wgu, &L.wdown }) m->weight_bytes += w->bytes;
            HIP_TRY(hipMalloc((void **) &L.k_cache, (size_t)(ML * KD) * 2));
            HIP_TRY(hipMalloc((void **) &L.v_cache, (size_t)(ML * KD) * 2));
            HIP_TRY(hipMemsetAsync(L.k_cache, 0, (size_t)(ML * KD) * 2, m->st));
            HIP_TRY(hipMemsetAsync(L.v_cache, 0, (size_t)(ML * KD) * 2, m->st));
        }
        HIP_TRY(hipMalloc((void **) &m->logits, (size_t) V * 4));
        HIP_TRY(hipMalloc((void **) &m->next_tok_dev, 16));
        HIP_TRY(hipMalloc((void **) &m->out_ring, (size_t) ML * 4));
        HIP_TRY(hipMalloc((void **) &m->counter_dev, (16 + 512 + 512) * 4));    // loop counter + 256 (value, index) argmax partials + cos/sin table of the position
        HIP_TRY(hipMalloc(&m->ffn_state, ffn_fused_state_bytes(F)));
        // the fused single-token path needs the row-concatenated projections and block-aligned widths
        m->fused_ok = m->own_stream && H % 256 == 0 && QD % 256 == 0 && hd % 8 == 0 && ML % 8 == 0 && (size_t)(hd + ML) * 4 <= 150 * 1024 &&
                      H <= 16384 && QD <= 32768 && F <= 32768 && F % 8 == 0;      // row lengths the decode mat-vec takes (gemv_decode.hip)
        for (const llama_layer & L : m->layers) {
            if (!L.wqkv.data || !L.wgu.data) m->fused_ok = false;
            else if (F % (L.wdown.type == CLLM_TYPE_Q4_K ? 256 : 32)) m->fused_ok = false;
            for (const dweight * w : { &L.wqkv, &L.wo, &L.wgu, &L.wdown }) if (w->data && !is_quant_type(w->type)) m->fused_ok = false;
        }
        if (!is_quant_type(m->lm_head.type)) m->fused_ok = false;
        m->finalized = true;
    }
    if (qlen > m->maxq) {
        if (m->decode_graph) { (void) hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
        if (m->decode_graph_long) { (void) hipGraphExecDestroy(m->decode_graph_long); m->decode_graph_long = nullptr; }
        HIP_TRY(hipStreamSynchronize(m->st));
        for (void * p : { (void *) m->x, (void *) m->xn, (void *) m->qkv, (void *) m->att, (void *) m->ctx, (void *) m->o, (void *) m->gu, (void *) m->g, m->wdata,
                          (void *) m->tokens_dev, (void *) m->pos_dev }) if (p) (void) hipFree(p);
        const size_t q = (size_t) qlen;
        HIP_TRY(hipMalloc((void **) &m->x,   q * H * 4));
        HIP_TRY(hipMalloc((void **) &m->xn,  q * H * 4));
        HIP_TRY(hipMalloc((void **) &m->qkv, q * (QD + 2*KD) * 4));
        HIP_TRY(hipMalloc((void **) &m->att, q * QD * 4));
        HIP_TRY(hipMalloc((void **) &m->ctx, q * QD * 4));
        HIP_TRY(hipMalloc((void **) &m->o,   q * H * 4));
        HIP_TRY(hipMalloc((void **) &m->gu,  q * 2 * F * 4));
        HIP_TRY(hipMalloc((void **) &m->g,   q * F * 4));
        int64_t kmax = H; if (F > kmax) kmax = F; if (QD > kmax) kmax = QD;
        m->wsize = act_row_bytes(kmax, 32) * q + 256;         // the Q8_0 kind is the larger of the two layouts
        HIP_TRY(hipMalloc(&m->wdata, m->wsize));
        HIP_TRY(hipMalloc((void **) &m->tokens_dev, q * 4));
        HIP_TRY(hipMalloc((void **) &m->pos_dev, q * 4));
        m->maxq = qlen;
    }
    return CLLM_OK;
}

static int ensure_scores(cllm_llama * m, size_t elems) {
    if (elems <= m->scores_elems) return CLLM_OK;
    HIP_TRY(hipStreamSynchronize(m->st));
    // the captured decode graphs hold the old pointer (long-context attention scratch): drop them, they are re-captured on demand
    if (m->decode_graph) { (void) hipGraphExecDestroy(m->decode_graph); m->decode_graph = nullptr; }
    if (m->decode_graph_long) { (void) hipGraphExecDestroy(m->decode_graph_long); m->decode_graph_long = nullptr; }
    if (m->scores) (void) hipFree(m->scores);
    m->scores = nullptr; m->scores_elems = 0;
    HIP_TRY(hipMalloc((void **) &m->scores, elems * 4));
    m->scores_elems = elems;
    return CLLM_OK;
}

// prefill (more than 32 columns): MUL_MAT with the SiLU*up quantizer prologue and / or the residual add in the epilogue (cllm_op_mul_mat_ex);
// CLLM_E_UNSUPPORTED: the caller issues the node sequence
static int linear_ex(cllm_llama * m, const dweight & w, int64_t K, int64_t N, float * x, int64_t qlen, float * y, int pro, float * resid, const float * norm_w = nullptr, int epi = 0) {
    if (!is_quant_type(w.type) || getenv("CLLM_NO_PREFILL_FUSE")) return CLLM_E_UNSUPPORTED;
    cllm_tensor W = T(w.type, w.data, K, N), X = T(CLLM_TYPE_F32, x, pro == 3 ? 2 * K : K, qlen), Y = T(CLLM_TYPE_F32, y, epi ? N / 2 : N, qlen), R = T(CLLM_TYPE_F32, resid, N, qlen);
    cllm_tensor G = T(CLLM_TYPE_F32, (void *) norm_w, K);
    return cllm_op_mul_mat_ex(m->st, &W, &X, &Y, m->wdata, m->wsize, pro, norm_w ? &G : nullptr, m->cfg.rms_eps, epi, resid ? &R : nullptr);
}
static int linear(cllm_llama * m, const dweight & w, int64_t K, int64_t N, float * x, int64_t qlen, float * y) {
    cllm_tensor W = T(w.type, w.data, K, N), X = T(CLLM_TYPE_F32, x, K, qlen), Y = T(CLLM_TYPE_F32, y, N, qlen);
    return cllm_op_mul_mat(m->st, &W, &X, &Y, m->wdata, m->wsize);
}

// ---- general path: one launch per graph node -----------------------------------------------------------------
static int forward_general(cllm_llama * m, int qlen, int n_past) {
    const cllm_llama_config & c = m->cfg;
    const int64_t H = c.hidden, hd = c.head_dim, nh = m->nh, nkv = m->nkv, QD = nh * hd, KD = nkv * hd, F = m->F, V = c.vocab, ML = c.max_len;
    const int64_t n_kv = (int64_t) n_past + qlen, QKV = QD + 2*KD;
    void * st = m->st;

    cllm_tensor ids = T(CLLM_TYPE_I32, m->tokens_dev, qlen), pos = T(CLLM_TYPE_I32, m->pos_dev, qlen);
    cllm_tensor X = T(CLLM_TYPE_F32, m->x, H, qlen), XN = T(CLLM_TYPE_F32, m->xn, H, qlen), O = T(CLLM_TYPE_F32, m->o, H, qlen);
    { cllm_tensor E = T(m->tok_embd.type, m->tok_embd.data, H, V); TRY(cllm_op_get_rows(st, &E, &ids, &X)); }
    cllm_rope_params rp = { (int32_t) hd, c.rope_mode, 0, c.rope_theta, 1.0f, 0.0f, 1.0f, 0.0f, 0.0f };

    for (int il = 0; il < c.n_layer; il++) {
        llama_layer & L = m->layers[il];
        cllm_tensor wn = T(CLLM_TYPE_F32, L.attn_norm.data, H);
        float * q = m->qkv, * k = m->qkv + QD, * v = m->qkv + QD + KD;     // row slices of the fused projection output
        int nrc = L.wqkv.data && H <= 16384 ? linear_ex(m, L.wqkv, H, QKV, m->x, qlen, m->qkv, 1, nullptr, (const float *) L.attn_norm.data) : CLLM_E_UNSUPPORTED;      // norm in the quantizer
        if (nrc != CLLM_OK && nrc != CLLM_E_UNSUPPORTED) return nrc;
        if (nrc == CLLM_E_UNSUPPORTED) TRY(cllm_op_rms_norm_mul(st, &X, &wn, &XN, c.rms_eps));
        if (nrc == CLLM_OK) {}
        else if (L.wqkv.data) TRY(linear(m, L.wqkv, H, QKV, m->xn, qlen, m->qkv));
        else {   // mixed-type q/k/v: three launches into strided slices
            for (int p = 0; p < 3; p++) {
                const dweight & w = p == 0 ? L.wq : p == 1 ? L.wk : L.wv; const int64_t N = p == 0 ? QD : KD;
                cllm_tensor W = T(w.type, w.data, H, N), Xn = T(CLLM_TYPE_F32, m->xn, H, qlen);
                cllm_tensor Y = TS(CLLM_TYPE_F32, p == 0 ? q : p == 1 ? k : v, N, qlen, 1, (size_t) QKV * 4, (size_t) QKV * 4 * qlen);
                TRY(cllm_op_mul_mat(st, &W, &Xn, &Y, m->wdata, m->wsize));
            }
        }
        if (c.qkv_bias) {
            for (int p = 0; p < 3; p++) {
                const dweight & b = p == 0 ? L.bq : p == 1 ? L.bk : L.bv; const int64_t N = p == 0 ? QD : KD;
                cllm_tensor Y = TS(CLLM_TYPE_F32, p == 0 ? q : p == 1 ? k : v, N, qlen, 1, (size_t) QKV * 4, (size_t) QKV * 4 * qlen);
                cllm_tensor B = T(CLLM_TYPE_F32, b.data, N);
                TRY(cllm_op_add(st, &Y, &B, &Y));
            }
        }
        // RoPE + the cache writes: one launch (the same bits as the four below); CLLM_NO_PREFILL_FUSE: the node sequence
        const bool rope_fused = !getenv("CLLM_NO_PREFILL_FUSE") && qlen > 1;
        if (rope_fused) TRY(launch_rope_kv_store((hipStream_t) st, m->qkv, QKV, m->pos_dev, qlen, (int) nh, (int) nkv, (int) hd, c.rope_mode, c.rope_theta, L.k_cache, L.v_cache, ML));
        else {
        // RoPE in place: k then q  ([hd, heads, qlen] views of the fused buffer)
        cllm_tensor Kt = TS(CLLM_TYPE_F32, k, hd, nkv, qlen, (size_t) hd * 4, (size_t) QKV * 4);
        cllm_tensor Qt = TS(CLLM_TYPE_F32, q, hd, nh,  qlen, (size_t) hd * 4, (size_t) QKV * 4);
        TRY(cllm_op_rope(st, &Kt, &pos, nullptr, &Kt, &rp));
        TRY(cllm_op_rope(st, &Qt, &pos, nullptr, &Qt, &rp));
        // KV-concat: K rows -> k_cache[pos] (SET_ROWS), V transposed -> v_cache[:, n_past..] (CPY into a strided view)
        {
            cllm_tensor Ks = TS(CLLM_TYPE_F32, k, KD, qlen, 1, (size_t) QKV * 4, (size_t) QKV * 4 * qlen);
            cllm_tensor Kc = T(CLLM_TYPE_F16, L.k_cache, KD, ML);
            TRY(cllm_op_set_rows(st, &Ks, &pos, &Kc));
            // src: transpose(v) = [qlen, KD] with nb0 = row stride of v; dst: view [qlen, KD] of v_cache with nb1 = 2*ML
            cllm_tensor Vt = T(CLLM_TYPE_F32, v, qlen, KD); Vt.nb[0] = (size_t) QKV * 4; Vt.nb[1] = 4; Vt.nb[2] = Vt.nb[3] = (size_t) QKV * 4 * qlen;
            cllm_tensor Vc = TS(CLLM_TYPE_F16, L.v_cache + n_past, qlen, KD, 1, (size_t) ML * 2, (size_t) ML * 2 * KD);
            TRY(cllm_op_cpy(st, &Vt, &Vc));
        }
        }
        // scores = K^T Q ; scale ; mask ; softmax ; ctx = V P
        {
            cllm_tensor Kv = TS(CLLM_TYPE_F16, L.k_cache, hd, n_kv, nkv, (size_t) KD * 2, (size_t) hd * 2);
            cllm_tensor Qv = TS(CLLM_TYPE_F32, q, hd, qlen, nh, (size_t) QKV * 4, (size_t) hd * 4);
            cllm_tensor Vv = TS(CLLM_TYPE_F16, L.v_cache, n_kv, hd, nkv, (size_t) ML * 2, (size_t) ML * hd * 2);
            int frc = CLLM_E_UNSUPPORTED;
            if (qlen >= flash_prefill_min_cols() && prefill_attn_mode() != 1) {     // fast mode, the tolerance tier (as the MFMA mat-muls below): one flash kernel, the scores never reach HBM
                tview vt = tv(&Vv); vt.ne[0] = ML;
                frc = launch_fattn((hipStream_t) st, tv(&Qv), tv(&Kv), CLLM_TYPE_F16, vt, 1, nullptr, n_past, (char *) m->att, (int64_t) QD * 4, (int64_t) hd * 4,
                                   (int64_t) QD * 4 * qlen, 1.0f / sqrtf((float) hd), nullptr, 0);
                if (frc != CLLM_OK && frc != CLLM_E_UNSUPPORTED) return frc;
            }
            if (frc == CLLM_E_UNSUPPORTED) {
            TRY(ensure_scores(m, (size_t)(n_kv * qlen * nh)));
            cllm_tensor S  = T(CLLM_TYPE_F32, m->scores, n_kv, qlen, nh);
            TRY(launch_mul_mat_f((hipStream_t) st, CLLM_TYPE_F16, tv(&Kv), tv(&Qv), tv(&S), 1, n_past));       // causal 1: fully masked tiles are not computed
            cllm_tensor C  = T(CLLM_TYPE_F32, m->ctx, hd, qlen, nh);
            // exact mode, a prompt: the probabilities stay fp16 between the soft-max and V.P (the rounding V.P's src1 conversion would apply: same bits, half the bytes)
            int prc = CLLM_E_UNSUPPORTED;
            if (prefill_attn_mode() == 1 && qlen > 32 && n_kv % 8 == 0 && hd % 4 == 0) {
                prc = launch_soft_max_causal_f16out((hipStream_t) st, tv(&S), 1.0f / sqrtf((float) hd), n_past);
                if (prc == CLLM_OK) { tview P = tv(&S); P.nb[0] = 2; prc = launch_mmf_exact((hipStream_t) st, tv(&Vv), P, tv(&C), 2, n_past, true); }
                if (prc != CLLM_OK && prc != CLLM_E_UNSUPPORTED) return prc;
            }
            if (prc == CLLM_E_UNSUPPORTED) {
            TRY(cllm_op_scale_mask_soft_max(st, &S, &S, 1.0f / sqrtf((float) hd), n_past));
            TRY(launch_mul_mat_f((hipStream_t) st, CLLM_TYPE_F16, tv(&Vv), tv(&S), tv(&C), 2, n_past));         // causal 2: k stops where P is exactly 0
            }
            // permute(0,2,1,3) + cont -> [hd, nh, qlen]
            cllm_tensor Cp = T(CLLM_TYPE_F32, m->ctx, hd, nh, qlen); Cp.nb[1] = (size_t) hd * qlen * 4; Cp.nb[2] = (size_t) hd * 4; Cp.nb[3] = (size_t) hd * qlen * nh * 4;
            cllm_tensor A  = T(CLLM_TYPE_F32, m->att, hd, nh, qlen);
            TRY(cllm_op_cpy(st, &Cp, &A));
            }
        }
        int orc = tp_on(m) ? CLLM_E_UNSUPPORTED : linear_ex(m, L.wo, QD, H, m->att, qlen, m->x, 0, m->x);      // x = wo . att + x in the GEMM's epilogue
        if (orc != CLLM_OK && orc != CLLM_E_UNSUPPORTED) return orc;
        if (orc == CLLM_E_UNSUPPORTED) {
        TRY(linear(m, L.wo, QD, H, m->att, qlen, m->o));
        if (tp_on(m)) TRY(tp_allreduce(m, (hipStream_t) st, m->o, H * qlen));
        TRY(cllm_op_add(st, &O, &X, &X));
        }

        cllm_tensor wf = T(CLLM_TYPE_F32, L.ffn_norm.data, H);
        int drc = CLLM_E_UNSUPPORTED;
        if (L.wgu.data && !tp_on(m) && H <= 16384) {       // norm in the gate/up quantizer; SiLU * up in down's quantizer (a memory-bound pass: the same work in the
            // gate/up GEMM's epilogue, epi 1, costs its VALU-bound main loop more than the smaller store saves: 131.5 vs 129.x ms); the residual add in down's epilogue
            drc = linear_ex(m, L.wgu, H, 2*F, m->x, qlen, m->gu, 1, nullptr, (const float *) L.ffn_norm.data);
            if (drc == CLLM_OK) drc = linear_ex(m, L.wdown, F, H, m->gu, qlen, m->x, 3, m->x);
            if (drc != CLLM_OK && drc != CLLM_E_UNSUPPORTED) return drc;
        }
        if (drc == CLLM_OK) continue;
        TRY(cllm_op_rms_norm_mul(st, &X, &wf, &XN, c.rms_eps));
        if (L.wgu.data) {
            TRY(linear(m, L.wgu, H, 2*F, m->xn, qlen, m->gu));
            if (!tp_on(m)) drc = linear_ex(m, L.wdown, F, H, m->gu, qlen, m->x, 3, m->x);      // x = wdown . (silu(gate) * up) + x: SiLU*up in the quantizer, the add in the epilogue
            if (drc != CLLM_OK && drc != CLLM_E_UNSUPPORTED) return drc;
        }
        if (drc == CLLM_OK) continue;
        if (L.wgu.data) {
            cllm_tensor G = TS(CLLM_TYPE_F32, m->gu, F, qlen, 1, (size_t) 2*F * 4, (size_t) 2*F * 4 * qlen);           // even elements
            cllm_tensor U = TS(CLLM_TYPE_F32, m->gu + 1, F, qlen, 1, (size_t) 2*F * 4, (size_t) 2*F * 4 * qlen);       // odd elements
            G.nb[0] = 8; U.nb[0] = 8;
            cllm_tensor Gd = T(CLLM_TYPE_F32, m->g, F, qlen);
            TRY(cllm_op_silu_mul(st, &G, &U, &Gd));
        } else {
            TRY(linear(m, L.wgate, H, F, m->xn, qlen, m->g));
            TRY(linear(m, L.wup, H, F, m->xn, qlen, m->gu));
            cllm_tensor G = T(CLLM_TYPE_F32, m->g, F, qlen), U = T(CLLM_TYPE_F32, m->gu, F, qlen);
            TRY(cllm_op_silu_mul(st, &G, &U, &G));
        }
        TRY(linear(m, L.wdown, F, H, m->g, qlen, m->o));
        if (tp_on(m)) TRY(tp_allreduce(m, (hipStream_t) st, m->o, H * qlen));
        TRY(cllm_op_add(st, &O, &X, &X));
    }
    // LMFinalSteps: last token -> norm -> lm_head
    cllm_tensor Xl = T(CLLM_TYPE_F32, m->x + (size_t)(qlen - 1) * H, H), XNl = T(CLLM_TYPE_F32, m->xn, H), wn = T(CLLM_TYPE_F32, m->out_norm.data, H);
    TRY(cllm_op_rms_norm_mul(st, &Xl, &wn, &XNl, c.rms_eps));
    TRY(linear(m, m->lm_head, H, V, m->xn, 1, m->logits));
    return CLLM_OK;
}

extern "C" int cllm_llama_forward(cllm_llama * m, const int32_t * tokens, int qlen, int n_past, float * logits_dev, float * logits_host) {
    if (!m || !tokens || qlen <= 0 || n_past < 0) FAIL(CLLM_E_INVALID, "llama_forward: arguments");
    if ((int64_t) n_past + qlen > m->cfg.max_len) FAIL(CLLM_E_INVALID, "llama_forward: context %d+%d exceeds max_len %d", n_past, qlen, m->cfg.max_len);
    TRY(finalize(m, qlen));
    std::vector<int32_t> pos(qlen);
    for (int i = 0; i < qlen; i++) { pos[i] = n_past + i; if (tokens[i] < 0 || tokens[i] >= m->cfg.vocab) FAIL(CLLM_E_INVALID, "llama_forward: token id out of range"); }
    HIP_TRY(hipMemcpyAsync(m->tokens_dev, tokens, (size_t) qlen * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos_dev, pos.data(), (size_t) qlen * 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));     // pos[] is a stack-lifetime host buffer
    TRY(forward_general(m, qlen, n_past));
    if (logits_dev) HIP_TRY(hipMemcpyAsync(logits_dev, m->logits, (size_t) m->cfg.vocab * 4, hipMemcpyDeviceToDevice, m->st));
    if (logits_host) { HIP_TRY(hipMemcpyAsync(logits_host, m->logits, (size_t) m->cfg.vocab * 4, hipMemcpyDeviceToHost, m->st)); HIP_TRY(hipStreamSynchronize(m->st)); }
    return CLLM_OK;
}

// ---- fused single-token step: 10 launches per layer, every per-token input read from device memory ----------------------

// cached positions above which the split attention (attn_long.hip: three launches, every CU) replaces the one-launch kernel (one CU per head).  Both accumulate in
// ggml_vec_dot_f16's order -- the same bits -- so the threshold is purely a speed matter: the measured crossover is ~400-500 cached positions.
int attn_long_threshold() { static const int v = getenv("CLLM_ATTN_LONG") ? atoi(getenv("CLLM_ATTN_LONG")) : 512; return v < 64 ? 64 : v; }

// the head of a step: the token's embedding row into m->x and the cos / sin table of its position.  The greedy loop runs it ONCE, in front of its first step: every sampled
// step ends with k_argmax_final_next, which prepares both for the step after it (CLLM_DECODE_FOLD=0: every step starts with its own head, as before round 5)
static bool decode_fold() { static const bool v = !(getenv("CLLM_DECODE_FOLD") && atoi(getenv("CLLM_DECODE_FOLD")) == 0); return v; }
static int decode_head(cllm_llama * m) {
    const cllm_llama_config & c = m->cfg;
    const int64_t H = c.hidden, hd = c.head_dim, V = c.vocab;
    cllm_tensor E = T(m->tok_embd.type, m->tok_embd.data, H, V), ids = T(CLLM_TYPE_I32, m->tokens_dev, 1), X = T(CLLM_TYPE_F32, m->x, H, 1);
    TRY(cllm_op_get_rows(m->st, &E, &ids, &X));
    // cos/sin of this step's position, once per token instead of once per head and layer (head sizes the compact kernel takes)
    if (hd == 64 || hd == 128) TRY(launch_rope_table(m->st, m->pos_dev, (int) hd, c.rope_theta, (float *)(m->counter_dev + 16 + 512)));
    return CLLM_OK;
}

// head == false (sampled steps of the greedy loop): x and the table were prepared by the previous step's last launch (or by decode_head in front of the loop)
static int decode_step_fused(cllm_llama * m, bool sample, bool long_ctx, bool head = true) {
    const cllm_llama_config & c = m->cfg;
    const int64_t H = c.hidden, hd = c.head_dim, QD = (int64_t) m->nh * hd, KD = (int64_t) m->nkv * hd, F = m->F, V = c.vocab, ML = c.max_len;
    hipStream_t st = m->st;
    const bool tp = tp_on(m);
    if (head) TRY(decode_head(m));
    float * rope_cs = (float *)(m->counter_dev + 16 + 512);
    const bool cs_table = (hd == 64 || hd == 128);
    // 5 launches per layer: [norm+quant+qkv GEMV(+bias)] [rope+kv-write+attention] [quant+o GEMV+residual]
    //                       [norm+quant+gate/up GEMV+silu*up] [quant+down GEMV+residual]
    // Tensor parallel: o / down write partial sums to m->o, all-reduced in place; the residual add x += o is folded into the NEXT
    // mat-vec's RMS_NORM prologue (which stores the new x into the other of two ping-pong buffers) when that kernel takes it,
    // else done by an ADD launch.
    float * xc = m->x;                       // current residual stream
    const float * pend = nullptr;            // all-reduced partial not yet added to xc
    // fused all-reduce (gemv_tp.hip): o / down scatter their partial rows into every rank's buffer (site 2 il / 2 il + 1), the next RMS_NORM mat-vec gathers them
    const bool fused_ar = tp && m->tp_fused != nullptr;
    const void * fctx = fused_ar ? cllm_tp_fused_dev(m->tp_fused) : nullptr;
    int pend_site = -1;
    if (fused_ar) TRY(cllm_tp_fused_advance(m->tp_fused, st));
    auto norm_gemv = [&](const dweight & w, int64_t nrows, const float * nw, int epi, float * dst, const float * bias) -> int {
        if (pend_site >= 0) {
            float * xo = xc == m->x ? m->xn : m->x;
            const int rc = launch_gemv_decode_tp_gather(st, w.type, w.data, H, nrows, xc, nw, c.rms_eps, epi, dst, bias, fctx, pend_site, xo);
            if (rc == CLLM_E_UNSUPPORTED) FAIL(CLLM_E_UNSUPPORTED, "decode: the fused all-reduce cannot take this mat-vec (type %d, %lld x %lld); unbind it (cllm_llama_set_tp_fused(m, NULL))", w.type, (long long) H, (long long) nrows);
            if (rc) return rc;
            xc = xo; pend_site = -1;
            return CLLM_OK;
        }
        if (pend) {
            float * xo = xc == m->x ? m->xn : m->x;
            int rc = launch_gemv_decode(st, w.type, w.data, H, nrows, 1, xc, nw, c.rms_eps, epi, dst, bias, nullptr, pend, xo);
            if (rc == CLLM_OK) { xc = xo; pend = nullptr; return rc; }
            if (rc != CLLM_E_UNSUPPORTED) return rc;
            cllm_tensor O = T(CLLM_TYPE_F32, (void *) pend, H), X = T(CLLM_TYPE_F32, xc, H);
            TRY(cllm_op_add(st, &O, &X, &X));
            pend = nullptr;
        }
        return launch_mmvq_fused(st, w.type, w.data, H, nrows, 1, xc, nw, c.rms_eps, epi, dst, bias, nullptr);
    };
    for (int il = 0; il < c.n_layer; il++) {
        llama_layer & L = m->layers[il];
        TRY(norm_gemv(L.wqkv, QD + 2*KD, (const float *) L.attn_norm.data, 0, m->qkv, c.qkv_bias ? (const float *) L.bqkv.data : nullptr));
        int arc = CLLM_E_UNSUPPORTED;
        if (cs_table && long_ctx) arc = launch_attn_long_flash(st, m->qkv, m->pos_dev, rope_cs, m->nh, m->nkv, (int) hd, c.rope_mode, L.k_cache, L.v_cache, ML, m->scores, m->scores_elems * 4, m->att);
        if (arc == CLLM_E_UNSUPPORTED && cs_table && long_ctx) arc = launch_attn_long(st, m->qkv, m->pos_dev, rope_cs, m->nh, m->nkv, (int) hd, c.rope_mode, L.k_cache, L.v_cache, ML, m->scores, m->att);
        if (arc == CLLM_E_UNSUPPORTED && cs_table) arc = launch_attn_dec_table(st, m->qkv, m->pos_dev, rope_cs, m->nh, m->nkv, (int) hd, c.rope_mode, L.k_cache, L.v_cache, ML, m->att);
        if (arc == CLLM_E_UNSUPPORTED) arc = launch_rope_kv_attn_decode(st, m->qkv, m->pos_dev, m->nh, m->nkv, (int) hd, c.rope_mode, c.rope_theta, L.k_cache, L.v_cache, ML, m->att);
        TRY(arc);
        if (!tp) TRY(launch_mmvq_fused(st, L.wo.type, L.wo.data, QD, H, 2, m->att, nullptr, 0.0f, 0, xc, nullptr, xc));          // x = o + x
        else if (fused_ar) {
            const int rc = launch_gemv_decode_tp_scatter(st, L.wo.type, L.wo.data, QD, H, 2, m->att, fctx, 2 * il);
            if (rc == CLLM_E_UNSUPPORTED) FAIL(CLLM_E_UNSUPPORTED, "decode: the fused all-reduce cannot take o_proj (type %d, K %lld); unbind it (cllm_llama_set_tp_fused(m, NULL))", L.wo.type, (long long) QD);
            if (rc) return rc;
            pend_site = 2 * il;
        } else {
            TRY(launch_mmvq_fused(st, L.wo.type, L.wo.data, QD, H, 2, m->att, nullptr, 0.0f, 0, m->o, nullptr, nullptr));
            TRY(tp_allreduce(m, st, m->o, H));
            pend = m->o;
        }
        // gate/up: the wave that owns a feature's row pair applies SiLU(gate)*up itself, the down mat-vec only quantizes;
        // otherwise the down mat-vec's prologue does SiLU*up on the interleaved pairs
        const bool silu_epi = F % 8 == 0 && H <= 16384;       // (every quantized type the decode kernel takes)
        const float * dsrc = silu_epi ? m->g : m->gu;
        const int dpro = silu_epi ? 2 : 3;
        // opt-in (CLLM_FFN_FUSED=1; bit-identical, slower: profiles/r06_ffn_fused_persistent_bound.txt): the whole FFN block as ONE launch where the shapes allow (ffn_fused.hip)
        if (ffn_fused_mode() && !tp && silu_epi && L.wgu.type == CLLM_TYPE_Q4_K && L.wdown.type == CLLM_TYPE_Q4_K) {
            const int frc = launch_ffn_fused(st, L.wgu.data, L.wdown.data, H, F, xc, (const float *) L.ffn_norm.data, c.rms_eps, m->ffn_state, &m->ffn_state_ready, xc);
            if (frc == CLLM_OK) continue;
            if (frc != CLLM_E_UNSUPPORTED) return frc;
        }
        TRY(norm_gemv(L.wgu, 2*F, (const float *) L.ffn_norm.data, silu_epi ? 1 : 0, silu_epi ? m->g : m->gu, nullptr));
        if (!tp) TRY(launch_mmvq_fused(st, L.wdown.type, L.wdown.data, F, H, dpro, dsrc, nullptr, 0.0f, 0, xc, nullptr, xc));   // x = down + x
        else if (fused_ar) {
            const int rc = launch_gemv_decode_tp_scatter(st, L.wdown.type, L.wdown.data, F, H, dpro, dsrc, fctx, 2 * il + 1);
            if (rc == CLLM_E_UNSUPPORTED) FAIL(CLLM_E_UNSUPPORTED, "decode: the fused all-reduce cannot take down_proj (type %d, K %lld); unbind it (cllm_llama_set_tp_fused(m, NULL))", L.wdown.type, (long long) F);
            if (rc) return rc;
            pend_site = 2 * il + 1;
        } else {
            TRY(launch_mmvq_fused(st, L.wdown.type, L.wdown.data, F, H, dpro, dsrc, nullptr, 0.0f, 0, m->o, nullptr, nullptr));
            TRY(tp_allreduce(m, st, m->o, H));
            pend = m->o;
        }
    }
    float * part_v = (float *)(m->counter_dev + 16); int * part_i = (int *)(m->counter_dev + 16 + 256);
    if (sample && !head) {
        // lm_head with the sampler's first stage in its epilogue (k_gemv_rows EPI 2: the values are still in registers) when that kernel takes the launch, then the second
        // stage + the next step's head as one launch
        int np = 256, rc = CLLM_E_UNSUPPORTED;
        if (pend_site < 0 && !pend && m->lm_head.type == CLLM_TYPE_Q4_K)
            rc = launch_gemv_rows_argmax(st, m->lm_head.data, H, V, xc, (const float *) m->out_norm.data, c.rms_eps, m->logits, part_v, part_i, &np);
        if (rc == CLLM_E_UNSUPPORTED) {
            TRY(norm_gemv(m->lm_head, V, (const float *) m->out_norm.data, 0, m->logits, nullptr));
            TRY(launch_argmax_partial(st, m->logits, (int) V, part_v, part_i));
            np = 256;
        } else if (rc) return rc;
        TRY(launch_argmax_final_next(st, part_v, part_i, np, m->tokens_dev, m->pos_dev, m->out_ring, m->counter_dev, m->tok_embd.type, m->tok_embd.data,
                                     cllm_row_size(m->tok_embd.type, H), (int) H, m->x, cs_table ? (int) hd : 0, c.rope_theta, rope_cs));
        return CLLM_OK;
    }
    TRY(norm_gemv(m->lm_head, V, (const float *) m->out_norm.data, 0, m->logits, nullptr));
    if (sample) TRY(launch_argmax_advance(st, m->logits, (int) V, m->tokens_dev, m->pos_dev, m->out_ring, m->counter_dev, part_v, part_i));
    return CLLM_OK;
}

// after a synchronize: did a bounded in-kernel wait time out?  (such a launch winds down instead of hanging the GPU; the step's results are void)
static int kernel_wait_check(cllm_llama * m) {
    TRY(gemv_team32_check());                                          // (the same kind of bounded wait inside a workgroup: gemv_team32.hip)
    // the one-shot all-reduce sums whatever its slots hold after a timed-out flag wait (a slow or dead peer): the step's results are void
    if (m->tp_oneshot && cllm_tp_oneshot_error(m->tp_oneshot)) FAIL(CLLM_E_HIP, "decode: a flag wait of the one-shot all-reduce timed out (a peer rank is late, dead or out of step): the step's logits are void");
    if (m->tp_fused && cllm_tp_fused_error(m->tp_fused)) FAIL(CLLM_E_HIP, "decode: a granule wait of the fused all-reduce timed out (a peer rank is late, dead or out of step): the step's logits are void");
    return CLLM_OK;
}

// capture one sampled step into a graph (after one eager warm-up step has set every function attribute)
static int ensure_decode_graph(cllm_llama * m, bool long_ctx) {
    hipGraphExec_t & slot = long_ctx ? m->decode_graph_long : m->decode_graph;
    if (slot || !m->use_graph || (m->cfg.tp_size > 1 && !m->tp_comm && !m->tp_oneshot && !m->tp_fused)) return CLLM_OK;     // a host callback cannot be captured; RCCL, the one-shot kernel and the fused form can
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamBeginCapture(m->st, hipStreamCaptureModeRelaxed));
    const int rc = decode_step_fused(m, true, long_ctx, !decode_fold());
    const hipError_t e = hipStreamEndCapture(m->st, &graph);
    hipError_t ei = hipSuccess;
    if (!rc && e == hipSuccess) ei = hipGraphInstantiate(&slot, graph, nullptr, nullptr, 0);
    if (graph) (void) hipGraphDestroy(graph);
    if (rc || e != hipSuccess || ei != hipSuccess) {
        // with a collective inside (tensor parallel) a failed capture is not fatal: the steps are launched eagerly instead
        if (m->cfg.tp_size > 1 && m->tp_comm) {
            (void) hipGetLastError();
            slot = nullptr; m->use_graph = false;
            fprintf(stderr, "[cllm] decode graph capture with RCCL inside failed (%s): launching the decode steps eagerly\n",
                    rc ? cllm_last_error() : hipGetErrorString(e != hipSuccess ? e : ei));
            return CLLM_OK;
        }
        if (rc) return rc;
        HIP_TRY(e);
        HIP_TRY(ei);
    }
    return CLLM_OK;
}

// ---- greedy decode loop (sampler = std::max_element, src/models.cpp:676-690: first maximum wins) ------------------
__global__ void __launch_bounds__(1024) k_argmax(const float * __restrict__ x, int n, int32_t * __restrict__ out) {
    __shared__ float bv[16]; __shared__ int bi[16];
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const float v = x[i]; if (v > best) { best = v; idx = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64); const int oi = __shfl_xor(idx, o, 64);
        if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
    }
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); w++) if (bv[w] > best || (bv[w] == best && bi[w] < idx)) { best = bv[w]; idx = bi[w]; }
        out[0] = idx == 0x7fffffff ? 0 : idx;
    }
}

extern "C" int cllm_llama_decode_greedy(cllm_llama * m, int32_t first_token, int n_past, int n_steps, int32_t * out_tokens_host) {
    if (!m || n_steps <= 0 || !out_tokens_host || n_past < 0) FAIL(CLLM_E_INVALID, "decode_greedy: arguments");
    if ((int64_t) n_past + n_steps > m->cfg.max_len) FAIL(CLLM_E_INVALID, "decode_greedy: exceeds max_len");
    if (first_token < 0 || first_token >= m->cfg.vocab) FAIL(CLLM_E_INVALID, "decode_greedy: token id out of range");
    TRY(finalize(m, 1));
    if (getenv("CLLM_DEBUG")) fprintf(stderr, "[cllm] decode_greedy: fused_ok=%d own_stream=%d use_graph=%d graph=%p\n", (int) m->fused_ok, (int) m->own_stream, (int) m->use_graph, (void *) m->decode_graph);
    if (m->fused_ok && !getenv("CLLM_NO_FUSED")) {
        const int32_t init[2] = { first_token, n_past }, zero = 0;
        HIP_TRY(hipMemcpyAsync(m->tokens_dev, &init[0], 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->pos_dev, &init[1], 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->counter_dev, &zero, 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        TRY(ensure_scores(m, (size_t) m->nh * m->cfg.max_len * 3 / 2 + 64));
        const int thr = attn_long_threshold();
        const bool fold = decode_fold();
        if (fold) TRY(decode_head(m));                             // the first step's head; every step prepares the next one's
        for (int s = 0; s < n_steps; s++) {
            const bool lng = n_past + s + 1 > thr;                 // cached positions this step attends to
            bool & warm = lng ? m->fused_warm_long : m->fused_warm;
            if (!warm) {                                           // one eager step sets every function attribute before the capture
                TRY(decode_step_fused(m, true, lng, !fold)); warm = true; HIP_TRY(hipStreamSynchronize(m->st));
                continue;
            }
            TRY(ensure_decode_graph(m, lng));
            hipGraphExec_t ge = lng ? m->decode_graph_long : m->decode_graph;
            if (ge) HIP_TRY(hipGraphLaunch(ge, m->st));
            else TRY(decode_step_fused(m, true, lng, !fold));
        }
        HIP_TRY(hipMemcpyAsync(out_tokens_host, m->out_ring, (size_t) n_steps * 4, hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        return kernel_wait_check(m);
    }
    int32_t tok = first_token;
    for (int s = 0; s < n_steps; s++) {
        const int32_t p = n_past + s;
        HIP_TRY(hipMemcpyAsync(m->tokens_dev, &tok, 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipMemcpyAsync(m->pos_dev, &p, 4, hipMemcpyHostToDevice, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        TRY(forward_general(m, 1, p));
        hipLaunchKernelGGL(k_argmax, dim3(1), dim3(1024), 0, m->st, m->logits, m->cfg.vocab, m->next_tok_dev);
        LAUNCH_CHECK();
        HIP_TRY(hipMemcpyAsync(&tok, m->next_tok_dev, 4, hipMemcpyDeviceToHost, m->st));
        HIP_TRY(hipStreamSynchronize(m->st));
        out_tokens_host[s] = tok;
    }
    return CLLM_OK;
}

/* one fused decode step WITHOUT sampling: logits of `token` at position n_past (for parity tests of the fused path) */
extern "C" int cllm_llama_decode_fused_logits(cllm_llama * m, int32_t token, int n_past, float * logits_host) {
    if (!m || !logits_host || n_past < 0 || n_past >= m->cfg.max_len || token < 0 || token >= m->cfg.vocab) FAIL(CLLM_E_INVALID, "decode_fused_logits: arguments");
    TRY(finalize(m, 1));
    if (!m->fused_ok) FAIL(CLLM_E_UNSUPPORTED, "decode_fused_logits: this model/config cannot take the fused path");
    const int32_t init[2] = { token, n_past };
    HIP_TRY(hipMemcpyAsync(m->tokens_dev, &init[0], 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipMemcpyAsync(m->pos_dev, &init[1], 4, hipMemcpyHostToDevice, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    TRY(ensure_scores(m, (size_t) m->nh * m->cfg.max_len * 3 / 2 + 64));
    TRY(decode_step_fused(m, false, n_past + 1 > attn_long_threshold()));
    HIP_TRY(hipMemcpyAsync(logits_host, m->logits, (size_t) m->cfg.vocab * 4, hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    return kernel_wait_check(m);
}

/* debug/test hook: copy an internal activation buffer of the LAST forward to the host ("x", "qkv", "att", "gu", "logits") */
extern "C" int cllm_llama_debug_read(cllm_llama * m, const char * what, float * host, int64_t n) {
    if (!m || !what || !host || n <= 0) FAIL(CLLM_E_INVALID, "debug_read: arguments");
    const std::string w(what);
    const float * src = w == "x" ? m->x : w == "qkv" ? m->qkv : w == "att" ? m->att : w == "gu" ? m->gu : w == "logits" ? m->logits : nullptr;
    if (!src) FAIL(CLLM_E_INVALID, "debug_read: unknown buffer '%s'", what);
    HIP_TRY(hipMemcpyAsync(host, src, (size_t) n * 4, hipMemcpyDeviceToHost, m->st));
    HIP_TRY(hipStreamSynchronize(m->st));
    return CLLM_OK;
}
