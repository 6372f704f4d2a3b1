/*
 * oracle/ref_ops.c -- TEST INFRASTRUCTURE.  Op-level harness over the REAL reference.
 *
 * Our own code (nothing copied): builds one-node (or few-node) graphs with the reference's
 * public ggml API (ggml/include/ggml.h, ggml-cpu.h) and runs them on the reference's CPU
 * backend (oracle/_ref/libggml-cpu.so, compiled from /root/reference by oracle/Makefile).
 * It is what pins oracle/ggml_oracle.c and what generates tests/golden/*.npz.
 * All inputs/outputs are dense host arrays; strided/permuted operands are produced the same way
 * chatllm does (ggml_view_* / ggml_permute on a dense parent), see ref_attention().
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

#include "ggml.h"
#include "ggml-cpu.h"

static int g_threads = 1;

void ref_set_threads(int n) { g_threads = n > 0 ? n : 1; }

static struct ggml_context * ctx_new(size_t bytes) {
    struct ggml_init_params p = { bytes + (64u << 20), NULL, false };
    return ggml_init(p);
}
static int run(struct ggml_context * ctx, struct ggml_tensor * out) {
    struct ggml_cgraph * gf = ggml_new_graph(ctx);
    ggml_build_forward_expand(gf, out);
    return ggml_graph_compute_with_ctx(ctx, gf, g_threads) == GGML_STATUS_SUCCESS ? 0 : -1;
}

/* ---- quantizers through the CPU type-traits tables (the exact functions mul_mat uses) ---- */
int ref_quantize_cpu(int type, const float * x, void * y, int64_t k) {    /* type_traits_cpu[type].from_float */
    ggml_cpu_init();
    const struct ggml_type_traits_cpu * t = ggml_get_type_traits_cpu((enum ggml_type) type);
    if (!t || !t->from_float) return -1;
    t->from_float(x, y, k);
    return 0;
}
int ref_quantize_ref(int type, const float * x, void * y, int64_t k) {    /* ggml_get_type_traits(type)->from_float_ref */
    const struct ggml_type_traits * t = ggml_get_type_traits((enum ggml_type) type);
    if (!t || !t->from_float_ref) return -1;
    t->from_float_ref(x, y, k);
    return 0;
}
int ref_dequantize(int type, const void * x, float * y, int64_t k) {
    const struct ggml_type_traits * t = ggml_get_type_traits((enum ggml_type) type);
    if (!t || !t->to_float) return -1;
    t->to_float(x, y, k);
    return 0;
}
/* one vec_dot call: s = <w row, a row>, a already in vec_dot_type */
int ref_vec_dot(int wtype, int64_t n, const void * w, const void * a, float * s) {
    ggml_cpu_init();   /* fills the fp16->fp32 lookup table the x86 vec_dots read (simd-mappings.h:133-140) */
    const struct ggml_type_traits_cpu * t = ggml_get_type_traits_cpu((enum ggml_type) wtype);
    if (!t || !t->vec_dot) return -1;
    t->vec_dot((int) n, s, 0, w, 0, a, 0, 1);
    return 0;
}
size_t ref_row_size(int type, int64_t ne) { return ggml_row_size((enum ggml_type) type, ne); }

/* ---- mul_mat: w [K, N, ne02, 1] (wtype) x  x [K, M, ne12, 1] f32 -> out [N, M, ne12, 1] ---- */
int ref_mul_mat(int wtype, int64_t K, int64_t N, int64_t M, int64_t ne02, int64_t ne12,
                const void * w, const float * x, float * out) {
    const size_t wbytes = ggml_row_size((enum ggml_type) wtype, K) * (size_t)(N * ne02);
    struct ggml_context * ctx = ctx_new(wbytes + (size_t)(K*M*ne12 + N*M*ne12) * 4 + (size_t)(K*M*ne12) * 8);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, (enum ggml_type) wtype, K, N, ne02);
    struct ggml_tensor * b = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, K, M, ne12);
    memcpy(a->data, w, wbytes);
    memcpy(b->data, x, (size_t)(K*M*ne12) * 4);
    struct ggml_tensor * c = ggml_mul_mat(ctx, a, b);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(N*M*ne12) * 4);
    ggml_free(ctx);
    return rc;
}

/* ---- mul_mat_id: as [K, N, E], b [K, nb1, T] f32, ids [U, T] i32 -> out [N, U, T] ---- */
int ref_mul_mat_id(int wtype, int64_t K, int64_t N, int64_t E, int64_t nb1, int64_t U, int64_t T,
                   const void * w, const float * x, const int32_t * ids, float * out) {
    const size_t wbytes = ggml_row_size((enum ggml_type) wtype, K) * (size_t)(N * E);
    struct ggml_context * ctx = ctx_new(wbytes + (size_t)(K*nb1*T*3 + N*U*T + U*T) * 4 + (1u << 20));
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, (enum ggml_type) wtype, K, N, E);
    struct ggml_tensor * b = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, K, nb1, T);
    struct ggml_tensor * i = ggml_new_tensor_2d(ctx, GGML_TYPE_I32, U, T);
    memcpy(a->data, w, wbytes);
    memcpy(b->data, x, (size_t)(K*nb1*T) * 4);
    memcpy(i->data, ids, (size_t)(U*T) * 4);
    struct ggml_tensor * c = ggml_mul_mat_id(ctx, a, b, i);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(N*U*T) * 4);
    ggml_free(ctx);
    return rc;
}

/* ---- unary-ish ops on a dense f32 [n0, n1, n2] tensor ---- */
enum { REF_RMS_NORM = 0, REF_SILU = 1, REF_SOFT_MAX = 2, REF_DIAG_MASK_INF = 3, REF_SCALE = 4 };
int ref_unary(int op, int64_t n0, int64_t n1, int64_t n2, const float * x, float * out, float fparam, int iparam) {
    struct ggml_context * ctx = ctx_new((size_t)(n0*n1*n2) * 12);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, n0, n1, n2);
    memcpy(a->data, x, (size_t)(n0*n1*n2) * 4);
    struct ggml_tensor * c = NULL;
    switch (op) {
        case REF_RMS_NORM:      c = ggml_rms_norm(ctx, a, fparam); break;
        case REF_SILU:          c = ggml_silu(ctx, a); break;
        case REF_SOFT_MAX:      c = ggml_soft_max(ctx, a); break;
        case REF_DIAG_MASK_INF: c = ggml_diag_mask_inf(ctx, a, iparam); break;
        case REF_SCALE:         c = ggml_scale(ctx, a, fparam); break;
        default: ggml_free(ctx); return -2;
    }
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(n0*n1*n2) * 4);
    ggml_free(ctx);
    return rc;
}

/* soft_max_ext with a dense mask [n0, n1] (f32 if mask_f16 == 0 else f16 bits) broadcast over n2 */
int ref_soft_max_ext(int64_t n0, int64_t n1, int64_t n2, const float * x, const void * mask, int mask_f16, float scale, float * out) {
    struct ggml_context * ctx = ctx_new((size_t)(n0*n1*n2) * 12 + (size_t)(n0*n1) * 4);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, n0, n1, n2);
    memcpy(a->data, x, (size_t)(n0*n1*n2) * 4);
    struct ggml_tensor * m = ggml_new_tensor_2d(ctx, mask_f16 ? GGML_TYPE_F16 : GGML_TYPE_F32, n0, n1);
    memcpy(m->data, mask, (size_t)(n0*n1) * (mask_f16 ? 2 : 4));
    struct ggml_tensor * c = ggml_soft_max_ext(ctx, a, m, scale, 0.0f);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(n0*n1*n2) * 4);
    ggml_free(ctx);
    return rc;
}

/* the router ops of a sparse-MoE block on a dense f32 [n0, n1, n2]: sum_rows -> f32 [1, n1, n2]; top_k -> i32 [k, n1, n2] */
int ref_sum_rows(int64_t n0, int64_t n1, int64_t n2, const float * x, float * out) {
    struct ggml_context * ctx = ctx_new((size_t)(n0*n1*n2) * 8);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, n0, n1, n2);
    memcpy(a->data, x, (size_t)(n0*n1*n2) * 4);
    struct ggml_tensor * c = ggml_sum_rows(ctx, a);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(n1*n2) * 4);
    ggml_free(ctx);
    return rc;
}
int ref_top_k(int64_t n0, int64_t n1, int64_t n2, const float * x, int k, int32_t * out) {
    struct ggml_context * ctx = ctx_new((size_t)(n0*n1*n2) * 12);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, n0, n1, n2);
    memcpy(a->data, x, (size_t)(n0*n1*n2) * 4);
    struct ggml_tensor * c = ggml_top_k(ctx, a, k);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(k*n1*n2) * 4);
    ggml_free(ctx);
    return rc;
}

/* binary with broadcast: a [n0,n1,n2], b [m0,m1,m2]; op 0 add, 1 mul, 2 div */
int ref_binary(int op, int64_t n0, int64_t n1, int64_t n2, const float * x, int64_t m0, int64_t m1, int64_t m2, const float * y, float * out) {
    struct ggml_context * ctx = ctx_new((size_t)(n0*n1*n2) * 12 + (size_t)(m0*m1*m2) * 4);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, n0, n1, n2);
    struct ggml_tensor * b = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, m0, m1, m2);
    memcpy(a->data, x, (size_t)(n0*n1*n2) * 4);
    memcpy(b->data, y, (size_t)(m0*m1*m2) * 4);
    struct ggml_tensor * c = op == 0 ? ggml_add(ctx, a, b) : op == 1 ? ggml_mul(ctx, a, b) : ggml_div(ctx, a, b);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(n0*n1*n2) * 4);
    ggml_free(ctx);
    return rc;
}

/* rope on x [hd, heads, qlen] f32 with pos[qlen]; ff may be NULL */
int ref_rope(int64_t hd, int64_t heads, int64_t qlen, const float * x, const int32_t * pos, const float * ff,
             int n_dims, int mode, int n_ctx_orig, float freq_base, float freq_scale, float ext_factor,
             float attn_factor, float beta_fast, float beta_slow, float * out) {
    struct ggml_context * ctx = ctx_new((size_t)(hd*heads*qlen) * 12 + (size_t) qlen * 4 + (size_t) hd * 4);
    if (!ctx) return -1;
    struct ggml_tensor * a = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, hd, heads, qlen);
    struct ggml_tensor * p = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, qlen);
    struct ggml_tensor * f = NULL;
    memcpy(a->data, x, (size_t)(hd*heads*qlen) * 4);
    memcpy(p->data, pos, (size_t) qlen * 4);
    if (ff) { f = ggml_new_tensor_1d(ctx, GGML_TYPE_F32, n_dims / 2); memcpy(f->data, ff, (size_t)(n_dims/2) * 4); }
    struct ggml_tensor * c = ggml_rope_ext(ctx, a, p, f, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(hd*heads*qlen) * 4);
    ggml_free(ctx);
    return rc;
}

/* set_rows: dst [n0, rows_dst] (dst_type f16/f32, pre-filled from dst_io) <- src [n0, n_src] f32 at idx (i32 or i64) */
int ref_set_rows(int dst_type, int64_t n0, int64_t rows_dst, int64_t n_src, const float * src, const void * idx, int idx_i64, void * dst_io) {
    const size_t dbytes = ggml_row_size((enum ggml_type) dst_type, n0) * (size_t) rows_dst;
    struct ggml_context * ctx = ctx_new(dbytes + (size_t)(n0*n_src) * 4 + (size_t) n_src * 8);
    if (!ctx) return -1;
    struct ggml_tensor * d = ggml_new_tensor_2d(ctx, (enum ggml_type) dst_type, n0, rows_dst);
    struct ggml_tensor * s = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, n0, n_src);
    struct ggml_tensor * i = ggml_new_tensor_1d(ctx, idx_i64 ? GGML_TYPE_I64 : GGML_TYPE_I32, n_src);
    memcpy(d->data, dst_io, dbytes);
    memcpy(s->data, src, (size_t)(n0*n_src) * 4);
    memcpy(i->data, idx, (size_t) n_src * (idx_i64 ? 8 : 4));
    struct ggml_tensor * c = ggml_set_rows(ctx, d, s, i);
    int rc = run(ctx, c);
    if (!rc) memcpy(dst_io, d->data, dbytes);
    ggml_free(ctx);
    return rc;
}

/* the eager V-cache write of chatllm (src/layers.cpp:3082-3093): v [KD, qlen] f32 -> transpose -> CPY into the
 * strided view [qlen, KD] (nb1 = 2*max_len, offset n_past*2) of v_cache [max_len, KD] f16 */
int ref_cpy_v_cache(int64_t KD, int64_t qlen, int64_t max_len, int64_t n_past, const float * v, uint16_t * v_cache_io) {
    struct ggml_context * ctx = ctx_new((size_t)(KD*max_len) * 2 + (size_t)(KD*qlen) * 4);
    if (!ctx) return -1;
    struct ggml_tensor * vc = ggml_new_tensor_2d(ctx, GGML_TYPE_F16, max_len, KD);
    struct ggml_tensor * vv = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, KD, qlen);
    memcpy(vc->data, v_cache_io, (size_t)(KD*max_len) * 2);
    memcpy(vv->data, v, (size_t)(KD*qlen) * 4);
    struct ggml_tensor * vt   = ggml_transpose(ctx, vv);
    struct ggml_tensor * view = ggml_view_2d(ctx, vc, qlen, KD, 2 * (size_t) max_len, (size_t) n_past * 2);
    struct ggml_tensor * c    = ggml_cpy(ctx, vt, view);
    int rc = run(ctx, c);
    if (!rc) memcpy(v_cache_io, vc->data, (size_t)(KD*max_len) * 2);
    ggml_free(ctx);
    return rc;
}

/* get_rows: table [n0, rows] (type) , ids [n] -> out [n0, n] f32 */
int ref_get_rows(int type, int64_t n0, int64_t rows, const void * table, int64_t n, const int32_t * ids, float * out) {
    const size_t tbytes = ggml_row_size((enum ggml_type) type, n0) * (size_t) rows;
    struct ggml_context * ctx = ctx_new(tbytes + (size_t)(n0*n) * 4 + (size_t) n * 4);
    if (!ctx) return -1;
    struct ggml_tensor * t = ggml_new_tensor_2d(ctx, (enum ggml_type) type, n0, rows);
    struct ggml_tensor * i = ggml_new_tensor_1d(ctx, GGML_TYPE_I32, n);
    memcpy(t->data, table, tbytes);
    memcpy(i->data, ids, (size_t) n * 4);
    struct ggml_tensor * c = ggml_get_rows(ctx, t, i);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(n0*n) * 4);
    ggml_free(ctx);
    return rc;
}

/* eager attention exactly as chatllm builds it (src/layers.cpp:3125-3179, 2541-2561, 2499-2539):
 *   q [hd, nh, qlen] f32 (post-rope), k_cache [KD, max_len] f16 rows = positions, v_cache [max_len, KD] f16 (transposed),
 *   scores = mul_mat(K view permuted, Q permuted) ; scale ; diag_mask_inf(n_past) ; soft_max ; mul_mat(V view, P) ;
 *   permute(0,2,1,3) ; cont  -> out [hd*nh, qlen] */
int ref_attention(int64_t hd, int64_t nh, int64_t nkv, int64_t qlen, int64_t n_past, int64_t max_len,
                  const float * q, const uint16_t * k_cache, const uint16_t * v_cache, float * out, float * scores_out) {
    const int64_t KD = hd * nkv, n_kv = n_past + qlen;
    struct ggml_context * ctx = ctx_new((size_t)(KD*max_len) * 4 + (size_t)(hd*nh*qlen) * 16 + (size_t)(n_kv*qlen*nh) * 24);
    if (!ctx) return -1;
    struct ggml_tensor * kc = ggml_new_tensor_1d(ctx, GGML_TYPE_F16, KD * max_len);
    struct ggml_tensor * vc = ggml_new_tensor_1d(ctx, GGML_TYPE_F16, KD * max_len);
    struct ggml_tensor * qq = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, hd, nh, qlen);
    memcpy(kc->data, k_cache, (size_t)(KD*max_len) * 2);
    memcpy(vc->data, v_cache, (size_t)(KD*max_len) * 2);
    memcpy(qq->data, q, (size_t)(hd*nh*qlen) * 4);

    struct ggml_tensor * query = ggml_permute(ctx, qq, 0, 2, 1, 3);                       /* [hd, qlen, nh] */
    struct ggml_tensor * key   = ggml_view_3d(ctx, kc, hd, nkv, n_kv, 2 * (size_t) hd, 2 * (size_t) KD, 0);
    key = ggml_permute(ctx, key, 0, 2, 1, 3);                                             /* [hd, n_kv, nkv] */
    struct ggml_tensor * value = ggml_view_3d(ctx, vc, n_kv, hd, nkv, 2 * (size_t) max_len, 2 * (size_t)(max_len * hd), 0);

    struct ggml_tensor * s = ggml_mul_mat(ctx, key, query);                               /* [n_kv, qlen, nh] */
    ggml_mul_mat_set_prec(s, GGML_PREC_F32);
    struct ggml_tensor * sraw = s;
    s = ggml_scale(ctx, s, 1.0f / sqrtf((float) hd));
    s = ggml_diag_mask_inf(ctx, s, (int) n_past);
    s = ggml_soft_max(ctx, s);
    struct ggml_tensor * c = ggml_mul_mat(ctx, value, s);                                 /* [hd, qlen, nh] */
    c = ggml_permute(ctx, c, 0, 2, 1, 3);
    c = ggml_cont(ctx, c);
    int rc = run(ctx, c);
    if (!rc) {
        memcpy(out, c->data, (size_t)(hd*nh*qlen) * 4);
        if (scores_out) memcpy(scores_out, sraw->data, (size_t)(n_kv*qlen*nh) * 4);
    }
    ggml_free(ctx);
    return rc;
}

/* flash attention as chatllm builds it with `-fa` (src/layers.cpp:2634-2656): q [D, N, H] f32 (already permuted), K / V caches
 * of `kv_type` (F16 | Q8_0 ...) given as dense rows [D, n_kv, Hkv], mask f16 [n_kv, N] or NULL -> out [D, H, N] */
int ref_flash_attn(int kv_type, int64_t D, int64_t N, int64_t H, int64_t Hkv, int64_t n_kv, const float * q, const void * k, const void * v,
                   const uint16_t * mask, float scale, float * out) {
    const size_t rb = ggml_row_size((enum ggml_type) kv_type, D);
    struct ggml_context * ctx = ctx_new((size_t)(D*N*H) * 16 + rb * (size_t)(n_kv*Hkv) * 2 + (size_t)(n_kv*N) * 2 + (64u << 20));
    if (!ctx) return -1;
    struct ggml_tensor * qq = ggml_new_tensor_3d(ctx, GGML_TYPE_F32, D, N, H);
    struct ggml_tensor * kk = ggml_new_tensor_3d(ctx, (enum ggml_type) kv_type, D, n_kv, Hkv);
    struct ggml_tensor * vv = ggml_new_tensor_3d(ctx, (enum ggml_type) kv_type, D, n_kv, Hkv);
    memcpy(qq->data, q, (size_t)(D*N*H) * 4);
    memcpy(kk->data, k, rb * (size_t)(n_kv*Hkv));
    memcpy(vv->data, v, rb * (size_t)(n_kv*Hkv));
    struct ggml_tensor * mm = NULL;
    if (mask) { mm = ggml_new_tensor_2d(ctx, GGML_TYPE_F16, n_kv, N); memcpy(mm->data, mask, (size_t)(n_kv*N) * 2); }
    struct ggml_tensor * c = ggml_flash_attn_ext(ctx, qq, kk, vv, mm, scale, 0.0f, 0.0f);
    int rc = run(ctx, c);
    if (!rc) memcpy(out, c->data, (size_t)(D*H*N) * 4);
    ggml_free(ctx);
    return rc;
}

/* ---- CPU baseline leg of bench.py: the reference's own mul_mat (all host threads), weights resident ----
 * w: n_copies distinct [K, N] matrices back to back (so the host LLC cannot hold the working set), x: [K] f32.
 * Runs `iters` single-token mat-vecs cycling through the copies; returns seconds per mat-vec. */
#include <time.h>
int ref_bench_mul_mat(int wtype, int64_t K, int64_t N, int n_copies, const void * w, const float * x, int iters, double * sec_per_iter) {
    const size_t wbytes = ggml_row_size((enum ggml_type) wtype, K) * (size_t) N;
    struct ggml_context * ctx = ctx_new(wbytes * (size_t) n_copies + (size_t)(K + N) * 4 * (size_t)(n_copies + 1) + (size_t) K * 8 * (size_t) n_copies);
    if (!ctx) return -1;
    struct ggml_tensor * b = ggml_new_tensor_2d(ctx, GGML_TYPE_F32, K, 1);
    memcpy(b->data, x, (size_t) K * 4);
    struct ggml_cgraph * gf[64];
    if (n_copies > 64) n_copies = 64;
    for (int c = 0; c < n_copies; c++) {
        struct ggml_tensor * a = ggml_new_tensor_2d(ctx, (enum ggml_type) wtype, K, N);
        memcpy(a->data, (const char *) w + wbytes * (size_t) c, wbytes);
        gf[c] = ggml_new_graph(ctx);
        ggml_build_forward_expand(gf[c], ggml_mul_mat(ctx, a, b));
    }
    ggml_graph_compute_with_ctx(ctx, gf[0], g_threads);                 /* warm-up: thread pool, work buffer */
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < iters; i++) ggml_graph_compute_with_ctx(ctx, gf[i % n_copies], g_threads);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    *sec_per_iter = ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec)) / (double) iters;
    ggml_free(ctx);
    return 0;
}
