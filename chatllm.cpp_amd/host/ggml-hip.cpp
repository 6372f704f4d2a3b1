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

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#define HIPB_LOG(...) do { fprintf(stderr, "[ggml-hip] " __VA_ARGS__); fputc('\n', stderr); } while (0)

namespace {

struct hip_device_ctx { int id; std::string name, desc; ggml_backend_buffer_type buft; };
struct hip_backend_ctx { int device; void * stream = nullptr; void * wdata = nullptr; size_t wsize = 0; };
struct hip_buffer_ctx { int device; void * base; };

std::vector<hip_device_ctx *> g_devices;
ggml_backend_reg g_reg;

cllm_tensor desc(const ggml_tensor * t) {
    cllm_tensor d;
    d.type = (int32_t) t->type;
    for (int i = 0; i < 4; i++) { d.ne[i] = t->ne[i]; d.nb[i] = t->nb[i]; }
    d.data = t->data;
    return d;
}
bool is_q(ggml_type t) { return t == GGML_TYPE_Q4_0 || t == GGML_TYPE_Q8_0 || t == GGML_TYPE_Q4_K; }
bool dense_rows(const ggml_tensor * t) { return t->nb[0] == ggml_type_size(t->type); }
bool f32_dense(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->nb[0] == 4; }

// ---------------------------------------------------------------------------------------------------------------------------
// buffer
// ---------------------------------------------------------------------------------------------------------------------------
void buf_free(ggml_backend_buffer_t b) { auto * c = (hip_buffer_ctx *) b->context; cllm_set_device(c->device); cllm_free(c->base); delete c; }
void * buf_base(ggml_backend_buffer_t b) { return ((hip_buffer_ctx *) b->context)->base; }
void buf_memset(ggml_backend_buffer_t b, ggml_tensor * t, uint8_t v, size_t off, size_t size) {
    cllm_set_device(((hip_buffer_ctx *) b->context)->device);
    cllm_memset((char *) t->data + off, v, size, nullptr); cllm_stream_sync(nullptr);
}
void buf_set(ggml_backend_buffer_t b, ggml_tensor * t, const void * data, size_t off, size_t size) {
    // called with 1,024,000-byte slices at arbitrary offsets while a model loads (src/chat.cpp:1322-1338): layout stays native
    cllm_set_device(((hip_buffer_ctx *) b->context)->device);
    cllm_memcpy_h2d((char *) t->data + off, data, size, nullptr); cllm_stream_sync(nullptr);
}
void buf_get(ggml_backend_buffer_t b, const ggml_tensor * t, void * data, size_t off, size_t size) {
    cllm_set_device(((hip_buffer_ctx *) b->context)->device);
    cllm_memcpy_d2h(data, (const char *) t->data + off, size, nullptr);
}
bool buf_cpy(ggml_backend_buffer_t b, const ggml_tensor * src, ggml_tensor * dst) {
    if (!ggml_is_contiguous(src) || !ggml_is_contiguous(dst) || ggml_nbytes(src) != ggml_nbytes(dst)) return false;
    cllm_set_device(((hip_buffer_ctx *) b->context)->device);
    if (ggml_backend_buffer_is_host(src->buffer)) { cllm_memcpy_h2d(dst->data, src->data, ggml_nbytes(src), nullptr); cllm_stream_sync(nullptr); return true; }
    if (src->buffer && src->buffer->iface.get_base == buf_base) {     // another buffer of this module (any device: peer access through hipMemcpy)
        cllm_memcpy_d2d(dst->data, src->data, ggml_nbytes(src), nullptr); cllm_stream_sync(nullptr); return true;
    }
    return false;
}
void buf_clear(ggml_backend_buffer_t b, uint8_t v) {
    auto * c = (hip_buffer_ctx *) b->context; cllm_set_device(c->device);
    cllm_memset(c->base, v, b->size, nullptr); cllm_stream_sync(nullptr);
}
const ggml_backend_buffer_i k_buffer_i = { buf_free, buf_base, nullptr, buf_memset, buf_set, buf_get, buf_cpy, buf_clear, nullptr };

// ---------------------------------------------------------------------------------------------------------------------------
// buffer type
// ---------------------------------------------------------------------------------------------------------------------------
const char * buft_name(ggml_backend_buffer_type_t t) { return ((hip_device_ctx *) t->context)->name.c_str(); }
ggml_backend_buffer_t buft_alloc(ggml_backend_buffer_type_t t, size_t size) {
    auto * d = (hip_device_ctx *) t->context;
    cllm_set_device(d->id);
    void * p = nullptr;
    if (cllm_malloc(&p, size ? size : 1) != CLLM_OK) { HIPB_LOG("alloc of %zu bytes failed: %s", size, cllm_last_error()); return nullptr; }
    return ggml_backend_buffer_init(t, k_buffer_i, new hip_buffer_ctx{ d->id, p }, size);
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
            if (is_q(a->type)) return a->ne[0] % 32 == 0 && (a->type != GGML_TYPE_Q4_K || (a->nb[1] % 16 == 0)) && b->nb[1] % 16 == 0;
            return a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32;
        case GGML_OP_MUL_MAT_ID:
            return is_q(a->type) && f32_dense(b) && op->src[2] && op->src[2]->type == GGML_TYPE_I32 && a->ne[3] == 1 && b->ne[3] == 1 &&
                   (b->ne[1] == 1 || b->ne[1] == op->src[2]->ne[0]) && op->src[2]->ne[0] * op->src[2]->ne[1] <= 65535;
        case GGML_OP_RMS_NORM: return f32_dense(a) && op->type == GGML_TYPE_F32;
        case GGML_OP_ADD: case GGML_OP_MUL: return a->type == GGML_TYPE_F32 && b->type == GGML_TYPE_F32 && op->type == GGML_TYPE_F32 && ggml_can_repeat(b, a);
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
        case GGML_OP_SET_ROWS: return f32_dense(a) && (op->type == GGML_TYPE_F16 || op->type == GGML_TYPE_F32) && (b->type == GGML_TYPE_I32 || b->type == GGML_TYPE_I64);
        case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: {
            const ggml_type s = a->type, d = op->type;
            const bool fs = s == GGML_TYPE_F32 || s == GGML_TYPE_F16, fd = d == GGML_TYPE_F32 || d == GGML_TYPE_F16;
            return (fs && fd) || (s == GGML_TYPE_I32 && d == GGML_TYPE_I32);
        }
        case GGML_OP_GET_ROWS: return (is_q(a->type) || a->type == GGML_TYPE_F16 || a->type == GGML_TYPE_F32) && b->type == GGML_TYPE_I32 && op->type == GGML_TYPE_F32;
        default: return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// backend (stream): graph_compute
// ---------------------------------------------------------------------------------------------------------------------------
const char * be_name(ggml_backend_t b) { return g_devices[((hip_backend_ctx *) b->context)->device]->name.c_str(); }
void be_free(ggml_backend_t b) {
    auto * c = (hip_backend_ctx *) b->context;
    cllm_set_device(c->device); cllm_stream_sync(c->stream);
    if (c->wdata) cllm_free(c->wdata);
    cllm_stream_destroy(c->stream);
    delete c; delete b;
}
void be_sync(ggml_backend_t b) { auto * c = (hip_backend_ctx *) b->context; cllm_set_device(c->device); cllm_stream_sync(c->stream); cllm_stream_sync(nullptr); }

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

// ---- node-pattern fusion (decode graphs: a token is ~25 launches per layer node by node, ~5 us each) -------------------------------
// A pattern is fused only if every intermediate it swallows is used by nobody else: the graph-wide use count
// (ggml_node_get_use_count, shared by scheduler splits) must equal the uses found in THIS graph, and the tensor must not be a
// graph output.  All fused launches are bit-identical to the node sequence they replace (same kernels' arithmetic).
//   RMS_NORM -> MUL(weight) -> {MUL_MAT ...}      norm + weight + quantize in every consumer mat-vec's prologue (pro 1)
//   UNARY(SILU) -> MUL(up) -> MUL_MAT             SiLU*up + quantize in the mat-vec's prologue (pro 4)
//   MUL_MAT -> ADD(residual | bias)               epilogue
//   SCALE -> DIAG_MASK_INF -> SOFT_MAX            cllm_op_scale_mask_soft_max
struct fused_mv { int pro = 2; const float * px = nullptr; const float * pw = nullptr; float eps = 0.0f; const float * resid = nullptr; float * dst = nullptr; };
struct fuse_plan {
    std::vector<uint8_t> skip;          // node is produced inside a fused launch (or not needed at all)
    std::vector<int>     mv;            // index into mvs for MUL_MAT nodes launched fused, else -1
    std::vector<fused_mv> mvs;
    std::vector<int>     sm_src;        // SOFT_MAX nodes: node index of the SCALE feeding the fused scale+mask+soft_max, else -1
};

bool f32_vec(const ggml_tensor * t) { return t && t->type == GGML_TYPE_F32 && t->nb[0] == 4 && t->ne[1] == 1 && t->ne[2] == 1 && t->ne[3] == 1 && ((uintptr_t) t->data & 15) == 0; }
bool overlap(const void * a, size_t na, const void * b, size_t nb) { return (const char *) a < (const char *) b + nb && (const char *) b < (const char *) a + na; }

fuse_plan make_plan(ggml_cgraph * g) {
    const int n = ggml_graph_n_nodes(g);
    fuse_plan P; P.skip.assign(n, 0); P.mv.assign(n, -1); P.sm_src.assign(n, -1);
    static const bool off = getenv("CLLM_HIP_NO_FUSE") != nullptr;
    if (off || n < 8) return P;
    std::vector<int> local(n, 0);
    std::vector<std::vector<int>> users(n);
    {   // uses inside this graph (views count as uses of their source, exactly like ggml's use counts)
        std::unordered_map<const ggml_tensor *, int> idx; idx.reserve((size_t) n * 2);
        for (int i = 0; i < n; i++) idx[ggml_graph_node(g, i)] = i;
        auto find = [&](const ggml_tensor * t) { auto it = idx.find(t); return it == idx.end() ? -1 : it->second; };
        for (int j = 0; j < n; j++) {
            const ggml_tensor * t = ggml_graph_node(g, j);
            for (int k = 0; k < GGML_MAX_SRC; k++) if (t->src[k]) { const int i = find(t->src[k]); if (i >= 0 && i < j) { local[i]++; users[i].push_back(j); } }
        }
    }
    auto only_local = [&](int i, int uses) {
        const ggml_tensor * t = ggml_graph_node(g, i);
        return !(t->flags & GGML_TENSOR_FLAG_OUTPUT) && local[i] == uses && ggml_node_get_use_count(g, i) == uses;
    };
    auto node_of = [&](const ggml_tensor * t, int before) { for (int i = before - 1; i >= 0 && i >= before - 64; i--) if (ggml_graph_node(g, i) == t) return i; return -1; };
    auto mv_ok = [&](const ggml_tensor * mm) {      // a single-column quantized MUL_MAT the decode mat-vec takes
        const ggml_tensor * w = mm->src[0], * x = mm->src[1];
        return mm->op == GGML_OP_MUL_MAT && is_q(w->type) && w->ne[2] == 1 && w->ne[3] == 1 && w->nb[1] == ggml_row_size(w->type, w->ne[0]) && ((uintptr_t) w->data & 15) == 0 &&
               x->type == GGML_TYPE_F32 && x->ne[1] == 1 && x->ne[2] == 1 && x->ne[3] == 1 && f32_vec(mm) &&
               w->ne[0] <= 16384 && (uint64_t) w->ne[1] * w->nb[1] < (1ull << 32);
    };
    for (int i = 0; i < n; i++) {
        ggml_tensor * t = ggml_graph_node(g, i);
        // ---- RMS_NORM -> MUL(weight) -> mat-vecs
        if (t->op == GGML_OP_MUL && t->src[0] && t->src[0]->op == GGML_OP_RMS_NORM && f32_vec(t) && f32_vec(t->src[0]->src[0]) && f32_vec(t->src[1]) &&
            t->src[1]->ne[0] == t->ne[0] && t->ne[0] % 256 == 0) {
            const int r = node_of(t->src[0], i);
            bool ok = r >= 0 && only_local(r, 1) && only_local(i, local[i]) && local[i] > 0;
            for (int j : users[i]) { const ggml_tensor * c = ggml_graph_node(g, j); ok = ok && mv_ok(c) && c->src[1] == t; }
            if (ok) {
                float eps; memcpy(&eps, t->src[0]->op_params, 4);
                for (int j : users[i]) {
                    fused_mv f; f.pro = 1; f.px = (const float *) t->src[0]->src[0]->data; f.pw = (const float *) t->src[1]->data; f.eps = eps; f.dst = (float *) ggml_graph_node(g, j)->data;
                    P.mv[j] = (int) P.mvs.size(); P.mvs.push_back(f);
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
                P.mv[j] = (int) P.mvs.size(); P.mvs.push_back(f);
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
            if (P.mv[j] < 0) { fused_mv f; f.pro = 2; f.px = (const float *) mm->src[1]->data; P.mv[j] = (int) P.mvs.size(); P.mvs.push_back(f); }
            fused_mv & f = P.mvs[P.mv[j]];
            const size_t kbytes = (size_t) mm->src[0]->ne[0] * 4, nbytes = (size_t) a->ne[0] * 4;
            if (overlap(a->data, nbytes, f.px, kbytes) || (f.pw && overlap(a->data, nbytes, f.pw, kbytes))) { if (f.pro == 2 && !f.resid && !f.dst) { P.mvs.pop_back(); P.mv[j] = -1; } continue; }
            f.resid = (const float *) r->data; f.dst = (float *) a->data;
            P.skip[i] = 1;
            break;
        }
    }
    for (int j = 0; j < n; j++) if (P.mv[j] >= 0 && !P.mvs[P.mv[j]].dst) P.mvs[P.mv[j]].dst = (float *) ggml_graph_node(g, j)->data;
    return P;
}

ggml_status be_graph_compute(ggml_backend_t backend, ggml_cgraph * g) {
    auto * c = (hip_backend_ctx *) backend->context;
    cllm_set_device(c->device);
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
    const fuse_plan plan = make_plan(g);
    for (int i = 0; i < ggml_graph_n_nodes(g); i++) {
        ggml_tensor * n = ggml_graph_node(g, i);
        if (ggml_is_empty(n) || plan.skip[i]) continue;
        const ggml_tensor * a = n->src[0], * b = n->src[1];
        int rc = CLLM_OK;
        cllm_tensor d = desc(n), da, db, dc;
        if (a) da = desc(a);
        if (b) db = desc(b);
        switch (n->op) {
            case GGML_OP_NONE: case GGML_OP_RESHAPE: case GGML_OP_VIEW: case GGML_OP_PERMUTE: case GGML_OP_TRANSPOSE: break;
            case GGML_OP_MUL_MAT: if (plan.mv[i] >= 0) {
                const fused_mv & f = plan.mvs[plan.mv[i]];
                rc = cllm_op_mul_mat_vec_fused(st, &da, f.pro, f.px, f.pw, f.eps, f.resid, f.dst);
            } else {
                const size_t need = cllm_mul_mat_wsize(&da, &db);
                if ((rc = ensure_wdata(c, need))) break;
                rc = cllm_op_mul_mat(st, &da, &db, &d, c->wdata, c->wsize);
            } break;
            case GGML_OP_MUL_MAT_ID: {
                dc = desc(n->src[2]);
                const size_t need = cllm_mul_mat_wsize(&da, &db);
                if ((rc = ensure_wdata(c, need))) break;
                rc = cllm_op_mul_mat_id(st, &da, &db, &dc, &d, c->wdata, c->wsize);
            } break;
            case GGML_OP_RMS_NORM: { float eps; memcpy(&eps, n->op_params, 4); rc = cllm_op_rms_norm(st, &da, &d, eps); } break;
            case GGML_OP_ADD: rc = cllm_op_add(st, &da, &db, &d); break;
            case GGML_OP_MUL: rc = cllm_op_mul(st, &da, &db, &d); break;
            case GGML_OP_SCALE: { float s, bias; memcpy(&s, n->op_params, 4); memcpy(&bias, (const float *) n->op_params + 1, 4); rc = cllm_op_scale(st, &da, &d, s, bias); } break;
            case GGML_OP_DIAG_MASK_INF: rc = cllm_op_diag_mask_inf(st, &da, &d, n->op_params[0]); break;
            case GGML_OP_UNARY: rc = cllm_op_unary(st, CLLM_UNARY_SILU, &da, &d); break;
            case GGML_OP_ROPE: {
                cllm_rope_params p;
                p.n_dims = n->op_params[1]; p.mode = n->op_params[2]; p.n_ctx_orig = n->op_params[4];
                memcpy(&p.freq_base, n->op_params + 5, 4); memcpy(&p.freq_scale, n->op_params + 6, 4); memcpy(&p.ext_factor, n->op_params + 7, 4);
                memcpy(&p.attn_factor, n->op_params + 8, 4); memcpy(&p.beta_fast, n->op_params + 9, 4); memcpy(&p.beta_slow, n->op_params + 10, 4);
                if (n->src[2]) dc = desc(n->src[2]);
                rc = cllm_op_rope(st, &da, &db, n->src[2] ? &dc : nullptr, &d, &p);
            } break;
            case GGML_OP_SOFT_MAX: if (plan.sm_src[i] >= 0) {
                const ggml_tensor * sc = ggml_graph_node(g, plan.sm_src[i]);
                float scale; memcpy(&scale, sc->op_params, 4);
                cllm_tensor ds = desc(sc->src[0]);
                rc = cllm_op_scale_mask_soft_max(st, &ds, &d, scale, n->src[0]->op_params[0]);
            } else {
                float scale, max_bias; memcpy(&scale, n->op_params, 4); memcpy(&max_bias, (const float *) n->op_params + 1, 4);
                rc = cllm_op_soft_max(st, &da, b ? &db : nullptr, &d, scale, max_bias);
            } break;
            case GGML_OP_SET_ROWS: rc = cllm_op_set_rows(st, &da, &db, &d); break;          // dst is a view of the cache (node->data)
            case GGML_OP_CPY: case GGML_OP_DUP: case GGML_OP_CONT: rc = cllm_op_cpy(st, &da, &d); break;
            case GGML_OP_GET_ROWS: rc = cllm_op_get_rows(st, &da, &db, &d); break;
            default: HIPB_LOG("graph_compute: op %s reached the device although supports_op() declined it", ggml_op_name(n->op)); return GGML_STATUS_FAILED;
        }
        if (rc != CLLM_OK) {
            HIPB_LOG("node %d (%s, '%s') failed: %s", i, ggml_op_name(n->op), n->name, cllm_last_error());
            return rc == CLLM_E_ALLOC ? GGML_STATUS_ALLOC_FAILED : GGML_STATUS_FAILED;
        }
    }
    return GGML_STATUS_SUCCESS;       // asynchronous: the host calls synchronize() before it reads (src/backend.cpp:824-825)
}

const ggml_backend_i k_backend_i = {
    be_name, be_free,
    nullptr, nullptr, nullptr,          // set/get_tensor_async, cpy_tensor_async: the scheduler falls back to the synchronous buffer calls
    be_sync,
    nullptr, nullptr, nullptr, nullptr, // graph plans
    be_graph_compute,
    nullptr, nullptr,                   // events
    nullptr,                            // graph_optimize
};
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
    p->caps = { /*async*/ true, /*host_buffer*/ false, /*buffer_from_host_ptr*/ false, /*events*/ false };
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
const ggml_backend_device_i k_device_i = {
    dev_name, dev_desc, dev_memory, dev_type, dev_props, dev_init, dev_buft,
    nullptr, nullptr,                   // host buffer type, buffer_from_host_ptr (SURVEY.md 8b note: keep NULL)
    dev_supports_op, dev_supports_buft,
    nullptr,                            // offload_op
    nullptr, nullptr, nullptr,          // events
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
    static bool done = false;
    if (!done) {
        done = true;
        const int n = cllm_device_count();
        g_dev_objs.reserve(n);
        for (int i = 0; i < n; i++) {
            auto * d = new hip_device_ctx();
            char name[256] = "MI355X"; int cus = 0;
            cllm_device_info(i, name, sizeof(name), nullptr, nullptr, &cus);
            d->id = i; d->name = "HIP" + std::to_string(i); d->desc = std::string(name) + ", " + std::to_string(cus) + " CUs (chatllm.cpp_amd)";
            g_devices.push_back(d);
            g_dev_objs.push_back(ggml_backend_device{ k_device_i, &g_reg, d });
            d->buft = ggml_backend_buffer_type{ k_buft_i, &g_dev_objs.back(), d };
        }
        g_reg = ggml_backend_reg{ GGML_BACKEND_API_VERSION, k_reg_i, nullptr };
    }
    return &g_reg;
}
int ggml_backend_score(void) { return cllm_device_count() > 0 ? 100 : 0; }
