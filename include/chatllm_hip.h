/*
 * chatllm_hip.h -- C ABI of the MI355X (gfx950) kernel library `libchatllm_hip.so`.
 *
 * This is the drop-in boundary for chatllm.cpp's transformer forward hot path.
 * Every entry point takes plain pointers / sizes / POD structs; no C++ or torch types.
 * All `data` pointers are DEVICE pointers; `stream` is a hipStream_t passed as void* (NULL = the
 * default stream).  Functions return 0 (CLLM_OK) or a negative cllm_status; they never throw
 * and never fall back to a CPU path: an unsupported type/shape is CLLM_E_UNSUPPORTED so the
 * caller (ggml's scheduler, via supports_op) can route the node elsewhere.
 *
 * Two callers bind it:
 *   1. chatllm.cpp_amd/host/ggml-hip.cpp -- a ggml backend module (ggml_backend_init() + the five vtables of
 *      /root/reference/ggml/src/ggml-backend-impl.h:11-251) whose graph_compute walks a ggml_cgraph
 *      and maps each node 1:1 onto the cllm_op_* functions below.  That is what the unmodified
 *      chatllm.cpp host (src/backend.cpp:277-302, --ggml_dir) loads as libggml-hip.so.
 *   2. chatllm.cpp_amd/ (Python, ctypes) -- tests and bench.
 *
 * The tensor descriptor mirrors struct ggml_tensor's {type, ne, nb, data}
 * (/root/reference/ggml/include/ggml.h:656-688): ne[0] is the contiguous (row) dimension,
 * nb[] are byte strides, quantized rows are arrays of blocks (ggml-common.h:170-175,219-224,295-306).
 */
#ifndef CHATLLM_HIP_H
#define CHATLLM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLLM_API __attribute__((visibility("default")))

typedef enum cllm_status {
    CLLM_OK            =  0,
    CLLM_E_INVALID     = -1,   /* bad shapes / null pointers              (ggml: GGML_ASSERT)            */
    CLLM_E_UNSUPPORTED = -2,   /* type/stride combination not implemented (ggml: supports_op == false)   */
    CLLM_E_HIP         = -3,   /* HIP runtime error, see cllm_last_error  (ggml: GGML_STATUS_FAILED)     */
    CLLM_E_ALLOC       = -4,   /* out of device memory                    (ggml: GGML_STATUS_ALLOC_FAILED) */
} cllm_status;

/* numeric values == enum ggml_type (ggml.h:386-428) so descriptors can be filled by a cast */
typedef enum cllm_type {
    CLLM_TYPE_F32 = 0, CLLM_TYPE_F16 = 1, CLLM_TYPE_Q4_0 = 2, CLLM_TYPE_Q4_1 = 3, CLLM_TYPE_Q8_0 = 8, CLLM_TYPE_Q4_K = 12, CLLM_TYPE_Q5_K = 13, CLLM_TYPE_Q6_K = 14,
    CLLM_TYPE_Q5_0 = 6, CLLM_TYPE_Q5_1 = 7, CLLM_TYPE_Q2_K = 10, CLLM_TYPE_Q3_K = 11, CLLM_TYPE_IQ4_NL = 20, CLLM_TYPE_IQ2_XXS = 16, CLLM_TYPE_IQ2_XS = 17, CLLM_TYPE_IQ3_XXS = 18, CLLM_TYPE_IQ1_S = 19, CLLM_TYPE_IQ3_S = 21, CLLM_TYPE_IQ2_S = 22, CLLM_TYPE_IQ4_XS = 23, CLLM_TYPE_IQ1_M = 29, CLLM_TYPE_TQ1_0 = 34, CLLM_TYPE_TQ2_0 = 35, CLLM_TYPE_MXFP4 = 39,     /* MUL_MAT (any columns) and GET_ROWS */
    CLLM_TYPE_I32 = 26, CLLM_TYPE_I64 = 27,
} cllm_type;

typedef struct cllm_tensor {
    int32_t type;        /* cllm_type */
    int64_t ne[4];       /* elements per dimension */
    size_t  nb[4];       /* byte strides */
    void *  data;        /* device pointer (already offset for views) */
} cllm_tensor;

/* ---- library / device ------------------------------------------------------------------- */
CLLM_API int          cllm_abi_version(void);                 /* == 1 */
CLLM_API int          cllm_device_count(void);
CLLM_API int          cllm_set_device(int device);
CLLM_API int          cllm_device_info(int device, char * name, size_t name_len, size_t * mem_free, size_t * mem_total,
                                       int * n_cu);           /* device_i.get_description/get_memory */
CLLM_API const char * cllm_last_error(void);                   /* thread-local message of the last failure */
CLLM_API size_t       cllm_type_size(int type);                /* ggml_type_size */
CLLM_API int          cllm_blck_size(int type);                /* ggml_blck_size */
CLLM_API size_t       cllm_row_size(int type, int64_t ne);     /* ggml_row_size  */

/* ---- memory / streams (buffer_i + backend_i plumbing: ggml-backend-impl.h:41-66, 87-127) --- */
CLLM_API int  cllm_malloc(void ** ptr, size_t size);           /* buffer_type_i.alloc_buffer  */
CLLM_API int  cllm_free(void * ptr);                           /* buffer_i.free_buffer        */
CLLM_API int  cllm_memset(void * dst, int value, size_t size, void * stream);            /* memset_tensor / clear */
CLLM_API int  cllm_memcpy_h2d(void * dst, const void * src, size_t size, void * stream); /* set_tensor[_async] */
CLLM_API int  cllm_memcpy_d2h(void * dst, const void * src, size_t size, void * stream); /* get_tensor[_async] */
CLLM_API int  cllm_memcpy_d2d(void * dst, const void * src, size_t size, void * stream); /* cpy_tensor        */
CLLM_API int  cllm_host_malloc(void ** ptr, size_t size);      /* page-locked host memory (staging for get_tensor; device_i.get_host_buffer_type) */
CLLM_API int  cllm_host_free(void * ptr);
CLLM_API int  cllm_stream_create(void ** stream);
CLLM_API int  cllm_stream_destroy(void * stream);
CLLM_API int  cllm_stream_sync(void * stream);                 /* backend_i.synchronize */
/* library-owned scratch of a stream the library did not create (a host's own stream; NULL = the legacy stream of the CURRENT device only): cllm_stream_destroy
 * never sees those.  Call with that stream idle. */
CLLM_API int  cllm_scratch_release(void * stream);
/* capture of everything launched on `stream` between begin and end into one replayable graph (ggml_backend_i.graph_plan_create /
 * graph_plan_compute, ggml-backend-impl.h:104-113); capture_end: CLLM_E_UNSUPPORTED and *graph_exec = NULL if the sequence cannot be captured */
/* after a synchronize: CLLM_E_HIP if a bounded in-kernel wait of the current device timed out since the last call (the launch wound down, its results are void) */
CLLM_API int  cllm_check_kernel_errors(void);
CLLM_API int  cllm_graph_capture_begin(void * stream);
CLLM_API int  cllm_graph_capture_end(void * stream, void ** graph_exec);
CLLM_API int  cllm_graph_launch(void * graph_exec, void * stream);
CLLM_API int  cllm_graph_destroy(void * graph_exec);
/* events on a stream (backend_i.event_record / device_i.event_synchronize); elapsed in milliseconds */
CLLM_API int  cllm_event_create(void ** event);
CLLM_API int  cllm_event_destroy(void * event);
CLLM_API int  cllm_event_record(void * event, void * stream);
CLLM_API int  cllm_event_sync(void * event);
CLLM_API int  cllm_event_elapsed_ms(void * start, void * stop, float * ms);
/* cross-stream / cross-device ordering for ggml_backend_i.event_wait and cpy_tensor_async (ggml-backend-impl.h:87-127; scheduler use:
 * ggml-backend.cpp:414-433, 1473-1477): `stream` waits for `event`; an asynchronous device-to-device copy between two devices of this
 * process, queued on `stream` (hipMemcpyPeerAsync; same device: a plain d2d copy). */
CLLM_API int  cllm_stream_wait_event(void * stream, void * event);
CLLM_API int  cllm_memcpy_peer_async(void * dst, int dst_device, const void * src, int src_device, size_t bytes, void * stream);

/* ---- the hot path: GGML_OP_MUL_MAT ------------------------------------------------------
 * replaces ggml_compute_forward_mul_mat (ggml/src/ggml-cpu/ggml-cpu.c:1229-1421) and, for Q4_0/Q8_0
 * with ne11 >= 2, llamafile_sgemm (ggml-cpu/llamafile/sgemm.cpp:3676-3696):
 *   dst[ne01, ne11, ne12, ne13] (F32) = src0^T . src1,   src0 broadcast over dims 2,3.
 * src0: Q4_0 | Q4_1 | Q8_0 | Q4_K (rows dense, any nb[1..3]) | F16 | F32 (any strides with nb[0]==elt size);
 * src1: F32.  Numerics follow the CPU path: src1 rows are quantized on the device to the weight
 * type's vec_dot_type (Q8_0: id=127/amax, round-half-even, arch/x86/quants.c:290-345;  Q8_1 for Q4_1: the
 * same plus s = fp16(d*sum q), :388-480;  Q8_K: ggml-quants.c:2555-2592;  F16: RNE), block dot products are
 * exact int32, scaling/accumulation fp32.
 * `wdata` is scratch for the converted src1 (ggml's params->wdata): at least cllm_mul_mat_wsize() bytes.
 */
CLLM_API size_t cllm_mul_mat_wsize(const cllm_tensor * src0, const cllm_tensor * src1);
CLLM_API int    cllm_op_mul_mat(void * stream, const cllm_tensor * src0, const cllm_tensor * src1, cllm_tensor * dst,
                                void * wdata, size_t wsize);

/* The prefill form of the node patterns around a quantized MUL_MAT (LMBlock1Forward / BaseMLP::forward, src/layers.cpp:2475-2483, 2719-2760):
 *   pro 0: MUL_MAT(src0, src1)
 *   pro 1: RMS_NORM(src1, eps) -> MUL(norm_w) -> MUL_MAT       (norm_w: dense F32 [K]; the normalised activation is never stored)
 *   pro 3: src1 holds 2 K interleaved (gate_e, up_e) pairs per row; the mat-mul runs over silu(gate) * up
 *   pro 4: UNARY(SILU)(src1) -> MUL(norm_w) -> MUL_MAT with gate = src1 and up = norm_w as separate F32 [K, M] tensors (the reference's own graph)
 *   pro 5: src1's act rows are already in wdata -- the previous cllm_op_mul_mat_ex / cllm_op_mul_mat on this stream quantized the SAME src1 (same K, M,
 *          weight block kind) and nothing else touched wdata since (several projections of one activation: q, k, v; gate, up); src1 gives the shape only
 *   epi 1: src0's rows alternate gate_u, up_u (cllm_pack_rows, interleave); dst F32 [N / 2, M] = silu(gate_u . x) * (up_u . x)   (MUL_MAT x 2 -> UNARY(SILU) -> MUL)
 *   resid != NULL (epi 0): ... -> ADD(resid)  (resid F32 of dst's shape; may be dst itself)
 * Only for src1->ne[1] >= cllm_mul_mat_ex_min_cols() (10 columns: where the exact-order GEMM beats the chunked mat-vec; CLLM_E_UNSUPPORTED below it: the caller issues the nodes).  The fused quantizers and
 * epilogues produce the bits of the separate RMS_NORM / MUL / SiLU / quantize / ADD passes. */
CLLM_API int    cllm_mul_mat_ex_min_cols(void);
/* How MUL_MAT with more than 32 activation columns and the prompt's attention block (cllm_op_attn_prefill, F16 MUL_MAT with > 32 columns) are computed:
 *   1 (default; CLLM_PREFILL=exact): the reference's accumulation ORDER for every prompt length -- the integer block sums of tinyBLAS_Q0_AVX / ggml_vec_dot_q4_K_q8_K
 *     (llamafile/sgemm.cpp:1346-1790, arch/x86/quants.c:1742-1822) on the K = 4 matrix-core instruction, their fp32 chains in block order; ggml_vec_dot_f16 /
 *     tinyBLAS<8> (vec.cpp:264-, sgemm.cpp:477-640) as fmaf chains on the f32 matrix cores: results bit-identical to libggml-cpu.so;
 *   0 (CLLM_PREFILL=fast): int8-MFMA GEMM + flash attention kernel, their own fp32 summation order (tolerance tier; the discontinuous activation quantizers
 *     downstream amplify that to ~0.1 sigma of the logits).
 * Process-wide; not meant to change while work is in flight. */
CLLM_API int    cllm_set_prefill_mode(int mode);
CLLM_API int    cllm_get_prefill_mode(void);
/* The prompt's attention block (K.Q -> soft_max -> V.P) on a switch of its own: -1 (default) follows the mode above, 0 the flash kernel, 1 the reference's order
 * (CLLM_PREFILL_ATTN=fast | exact).  "CLLM_PREFILL=fast CLLM_PREFILL_ATTN=exact" = int8-MFMA mat-muls + exact attention (profiles/r06_prefill_mode_decomposition.txt). */
CLLM_API int    cllm_set_prefill_attn_mode(int mode);
CLLM_API int    cllm_op_mul_mat_ex(void * stream, const cllm_tensor * src0, const cllm_tensor * src1, cllm_tensor * dst, void * wdata, size_t wsize,
                                   int pro, const cllm_tensor * norm_w, float eps, int epi, const cllm_tensor * resid);

/* OPT-IN tolerance tier of the single-column mat-vec for the 32-weight block formats (Q4_0 / Q4_1 / Q8_0; gemv_free32.hip; also CLLM_DECODE_FREE_ORDER=1): the exact int32 block
 * dot products of ggml_vec_dot_q4_0_q8_0 / q4_1_q8_1 / q8_0_q8_0 (ggml-cpu/arch/x86/quants.c:543-577, 701-760, 1012-1040) folded in a FREE fp32 order (per lane, then across the
 * wave) instead of the reference's eight per-AVX-lane chains.  Default 0: every decode mat-vec is bit-identical to the reference's CPU build. */
CLLM_API int    cllm_set_decode_free_order(int on);
CLLM_API int    cllm_get_decode_free_order(void);

/* The FFN block of ONE token (BaseMLP::forward src/layers.cpp:2475-2497 behind LMBlock1Forward's post_attention_layernorm and residual add :2744-2758) as ONE launch:
 *     xout = Wdown . q8_K(SiLU(Wgate . a) * (Wup . a)) + x,   a = q8_K(RMS_NORM(x, eps) * norm_w)
 * i.e. the nodes RMS_NORM -> MUL -> MUL_MAT x 2 -> UNARY(SILU) -> MUL -> MUL_MAT -> ADD, bit for bit (ggml_compute_forward_mul_mat ggml-cpu.c:1229-1421 per mat-mul).
 * w_gate_up: Q4_K [H, 2 F], rows alternate gate_u, up_u (cllm_pack_rows, interleave); w_down: Q4_K [F, H]; x, norm_w F32 [H]; xout may be x.
 * state: cllm_ffn_fused_state_bytes(F) bytes of device memory, zero-filled before the first call, private to one stream (the launch's hand-off granules and epoch).
 * CLLM_E_UNSUPPORTED (nothing launched) outside its shapes (H <= 4096, 4096 <= F, multiples of 256, a 256-CU device): issue the two cllm_op_mul_mat_vec_fused calls. */
CLLM_API size_t cllm_ffn_fused_state_bytes(int64_t F);
CLLM_API int    cllm_op_ffn_fused(void * stream, const cllm_tensor * w_gate_up, const cllm_tensor * w_down, const float * x, const float * norm_w, float eps, void * state, float * xout);

/* One launch for a node pattern around a single-column quantized MUL_MAT -- what a ggml backend's graph_compute can fuse
 * (chatllm.cpp_amd/host/ggml-hip.cpp does, with ggml's use counts):
 *   pro 1: RMS_NORM(px, eps) -> MUL(pw) -> MUL_MAT(src0)      pro 2: MUL_MAT(src0, px)      pro 4: UNARY(SILU)(px) -> MUL(pw) -> MUL_MAT(src0)
 *   resid != NULL: ... -> ADD(resid).   px / pw / resid / dst: dense F32 vectors (dst may alias resid, never px / pw).
 *   epi 1 (pro 1 only, no resid): src0 holds the gate and up projections of BaseMLP::forward (src/layers.cpp:2475-2483) with their rows
 *   alternating (cllm_pack_rows, interleave) and dst[i] = silu(row 2i . x) * (row 2i+1 . x): MUL_MAT, MUL_MAT, UNARY(SILU), MUL in one launch.
 * Bit-identical to the unfused cllm_op_* sequence.  CLLM_E_UNSUPPORTED for row lengths outside the decode kernel's range. */
CLLM_API int    cllm_op_mul_mat_vec_fused(void * stream, const cllm_tensor * src0, int pro, const float * px, const float * pw, float eps, int epi,
                                          const float * resid, float * dst);
/* device-side repack of weight rows so that mat-vecs reading the same activation become one launch: dst = srcs[0] rows | srcs[1] rows | ...
 * (interleave 0; e.g. q, k, v) or a_0, b_0, a_1, b_1, ... of two matrices with the same number of rows (interleave 1; gate, up). */
CLLM_API int    cllm_pack_rows(void * stream, void * dst, const void * const * srcs, const int64_t * nrows, int n, size_t row_bytes, int interleave);

/* gate and up expert mat-vecs of ONE token + UNARY(SILU) + MUL (MultiMLP::forward, src/layers.cpp:3674-3688) in one launch.
 * as_gu: [K, 2F, E], every expert's gate and up rows alternating (cllm_pack_rows over the two [K, F*E] tensors, interleave 1);
 * dst[u, slot] = silu(gate_e[u] . x) * (up_e[u] . x) with e = ids[slot] read on the device.  Bit-identical to the four nodes. */
CLLM_API int    cllm_op_mul_mat_id_silu_mul(void * stream, const cllm_tensor * as_gu, const cllm_tensor * b, const cllm_tensor * ids, cllm_tensor * dst);

/* measurement hook for bench.py's "roofline" object: quantizes src1 once, then times `iters` launches of ONLY the
 * mat-mul kernel between two HIP events on `stream`, cycling src0->data through src0_datas[0..n_src0) (distinct
 * copies of the weights, so the Infinity Cache cannot serve them).  avg_us = average kernel launch duration. */
CLLM_API int    cllm_bench_mul_mat_kernel(void * stream, const cllm_tensor * src0, void * const * src0_datas, int n_src0,
                                          const cllm_tensor * src1, cllm_tensor * dst, void * wdata, size_t wsize, int iters, float * avg_us);

/* same for GGML_OP_MUL_MAT_ID: `iters` launches of the whole op (activation quantization + expert mat-vecs), the ids cycled through
 * ids_datas[0..n_ids) (device pointers laid out like ids->data) so that successive launches pick different experts */
CLLM_API int    cllm_bench_mul_mat_id(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids, void * const * ids_datas, int n_ids,
                                      cllm_tensor * dst, void * wdata, size_t wsize, int iters, float * avg_us);

/* same, for the decode form of the mat-vec (single column, the activation is produced inside the kernel):
 *   pro 1: act = quantize(rms_norm(px[0..K)) * pw)   pro 2: act = quantize(px)   pro 3: act = quantize(silu(px[2i]) * px[2i+1])
 * dst[r] = W[r] . act (+ resid[r]);  epi 1: W rows alternate gate_u, up_u and dst[u] = silu(W[2u].act) * (W[2u+1].act).
 * These are the launches cllm_llama_decode_* issues per layer. */
/* measurement helper: GB/s of a pure streaming read of `bytes` (> the 256 MiB Infinity Cache) on the current device -- the achieved-read ceiling bench.py prints next to the nominal 8 TB/s */
CLLM_API int    cllm_bench_read_bw(void * stream, size_t bytes, int iters, float * gb_per_s);
CLLM_API int    cllm_bench_gemv_fused(void * stream, int wtype, void * const * w_datas, int n_w, int64_t K, int64_t nrows, int pro,
                                      const float * px, const float * pw, float eps, int epi, float * dst, const float * resid, int iters, float * avg_us);
/* measurement helper (tools/ffn_bench.py): the FFN block of a token `iters` times over n_w copies of its weights -- fused 0: the two launches of the five-launch layer,
 * 1: cllm_op_ffn_fused's launch, 2: the same with a hand-off that costs nothing (NOT a correct computation: the design's bound); x is updated in place */
CLLM_API int    cllm_bench_ffn(void * stream, void * const * w_gate_up, void * const * w_down, int n_w, int64_t H, int64_t F, float * x, const float * norm_w, float eps,
                               float * g, void * state, int fused, int iters, float * avg_us);

/* GGML_OP_MUL_MAT_ID -- ggml_compute_forward_mul_mat_id (ggml-cpu.c:1432-1678), chatllm MultiLinear::forward
 * (src/layers.cpp:2145-2151):  dst[:, s, t] = as[:, :, ids[s,t]]^T . b[:, s % b.ne1, t]              */
CLLM_API int    cllm_op_mul_mat_id(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids,
                                   cllm_tensor * dst, void * wdata, size_t wsize);

/* the activation quantizers on their own (KAT surface).  Output is in the reference block layout
 * (block_q8_0 34 B / block_q8_K 292 B) so it can be compared byte-for-byte. */
CLLM_API int    cllm_quantize_row_q8_0(void * stream, const float * x, void * y_blocks, int64_t k);   /* arch/x86/quants.c:290 */
CLLM_API int    cllm_quantize_row_q8_1(void * stream, const float * x, void * y_blocks, int64_t k);   /* arch/x86/quants.c:388 */
CLLM_API int    cllm_quantize_row_q8_K(void * stream, const float * x, void * y_blocks, int64_t k);   /* ggml-quants.c:2555    */
/* exact per-block integer sums of one weight row against one quantized activation row (tier T0).
 * Q4_0/Q4_1/Q8_0: one int32 per 32-block; Q4_K: {sum_s sc_s*dot_s, sum_s m_s*bsum_s} per super-block. */
CLLM_API int    cllm_vec_dot_isums(void * stream, int wtype, int64_t k, const void * w_row, const float * x, int32_t * isums);

/* ---- the other nodes of one forward graph -------------------------------- */
/* GGML_OP_RMS_NORM      ggml_compute_forward_rms_norm_f32  (ggml-cpu/ops.cpp:3710-3759) */
CLLM_API int cllm_op_rms_norm(void * stream, const cllm_tensor * src, cllm_tensor * dst, float eps);
/* fused RMS_NORM + MUL(weight) -- chatllm RMSNorm::forward (src/layers.cpp:2216-2225); same two roundings */
CLLM_API int cllm_op_rms_norm_mul(void * stream, const cllm_tensor * src, const cllm_tensor * weight, cllm_tensor * dst, float eps);

typedef struct cllm_rope_params {      /* op_params of GGML_OP_ROPE (ggml.c ggml_rope_impl) */
    int32_t n_dims, mode, n_ctx_orig;  /* mode 0 = NORMAL (i,i+1), 2 = NEOX (i, i+n_dims/2) */
    float   freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow;
} cllm_rope_params;
/* GGML_OP_ROPE          ggml_compute_forward_rope_flt<float> (ops.cpp:5589-5865); dst may alias src (rope_ext_inplace) */
CLLM_API int cllm_op_rope(void * stream, const cllm_tensor * src, const cllm_tensor * pos, const cllm_tensor * freq_factors,
                          cllm_tensor * dst, const cllm_rope_params * p);
/* GGML_OP_SOFT_MAX      ggml_compute_forward_soft_max_f32 (ops.cpp:5225-5335); mask may be NULL, F32 or F16 */
CLLM_API int cllm_op_soft_max(void * stream, const cllm_tensor * src, const cllm_tensor * mask, cllm_tensor * dst,
                              float scale, float max_bias);
/* GGML_OP_DIAG_MASK_INF ggml_compute_forward_diag_mask_f32 (ops.cpp:5137-5185) */
CLLM_API int cllm_op_diag_mask_inf(void * stream, const cllm_tensor * src, cllm_tensor * dst, int n_past);
/* GGML_OP_SCALE         ggml_compute_forward_scale (ops.cpp:4426-) y = x*s + b */
CLLM_API int cllm_op_scale(void * stream, const cllm_tensor * src, cllm_tensor * dst, float s, float b);
/* fused SCALE + DIAG_MASK_INF + SOFT_MAX as chatllm's attn_scores_to_probs emits them (src/layers.cpp:2499-2539) */
CLLM_API int cllm_op_scale_mask_soft_max(void * stream, const cllm_tensor * src, cllm_tensor * dst, float scale, int n_past);

/* GGML_OP_FLASH_ATTN_EXT  ggml_compute_forward_flash_attn_ext_f16 (ggml-cpu/ops.cpp:8114-8344, tiled :8346-8640, split-KV :8642-8770),
 * as CoreAttention emits it with `-fa` (src/layers.cpp:2634-2656):  dst[:, h, n] = soft_max_kv(scale * K.q + mask) . V
 * q F32 [D, N, H, B] (rows dense); k, v: F16 or Q8_0 [D, n_kv, Hkv, B] rows (--cache_dtype, src/layers.cpp:2925-2945); mask NULL or
 * F16 [n_kv, >= N, 1|H, 1|B]; dst F32 [D, H, N, B] contiguous.  D = 64 | 128, H % Hkv == 0, max_bias == 0 and logit_softcap == 0
 * (anything else: CLLM_E_UNSUPPORTED, the host keeps the node on the CPU).  TOLERANCE tier: Q is converted to K's vec_dot_type as
 * on the CPU (fp16, or quantize_row_q8_0), scores and the online soft-max are fp32, P is rounded to fp16 for the P.V product; the
 * CPU op itself changes summation order with shape and thread count.  wdata: cllm_flash_attn_wsize() bytes (split-KV partials). */
CLLM_API size_t cllm_flash_attn_wsize(const cllm_tensor * q);
CLLM_API int    cllm_op_flash_attn_ext(void * stream, const cllm_tensor * q, const cllm_tensor * k, const cllm_tensor * v, const cllm_tensor * mask,
                                       cllm_tensor * dst, float scale, float max_bias, float logit_softcap, void * wdata, size_t wsize);
/* the prefill attention block of the eager path, MUL_MAT(K, Q) + SCALE + DIAG_MASK_INF + SOFT_MAX + MUL_MAT(V^T, P) (src/layers.cpp:2499-2561),
 * in one call (qlen > 32).  Prefill mode 1 (default): K.Q -> soft_max -> V.P on the exact-order kernels (bit-identical to the node sequence and to the CPU; the
 * scores pass through a library-owned scratch buffer); mode 0: ONE flash kernel (tolerance tier).  q F32 [D, N, H]; k F16 [D, n_kv, Hkv] rows;
 * vt F16 [n_kv, D, Hkv] (the transposed V cache view); dst F32 [D, N, H] (any 16-byte aligned strides); causal with n_past. */
CLLM_API int    cllm_attn_prefill_min_cols(void);   /* query rows from which callers should use it (33; CLLM_MMA_MIN_COLS; CLLM_FLASH_PREFILL=0: never) */
CLLM_API int    cllm_op_attn_prefill(void * stream, const cllm_tensor * q, const cllm_tensor * k, const cllm_tensor * vt, cllm_tensor * dst,
                                     float scale, int n_past);

/* Greedy decode-ahead support for a host-side caller that replays a captured step: tok = index of the first maximum of logits[n]
 * (std::max_element, the greedy sampler of src/models.cpp) written to *tok_dev (device) and *tok_host (page-locked host memory the device can
 * write: cllm_host_malloc), and every int32 the n_inc device pointers of inc_ptrs_dev[] point at is incremented by one (the positions of the
 * next step).  scratch: 2 KB of device memory. */
CLLM_API int cllm_op_argmax_advance(void * stream, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host,
                                    int32_t * const * inc_ptrs_dev, int n_inc, void * scratch);
/* the same with ABSOLUTE values instead of increments: set_table_dev[] = n_set records { int32_t * ptr; int32_t val; int32_t pad; } (16 bytes each, device memory);
 * *ptr = val for every record (distinct pointers).  What the module's decode-ahead uses: the memory of a per-token scalar may have been reused by later nodes
 * of the previous graph, so its old content is not something to increment. */
CLLM_API int cllm_op_argmax_set(void * stream, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host, const void * set_table_dev, int n_set, void * scratch);
/* ... and with the snapshot of the graph's outputs in the same launch (what stood in front of every token through the unmodified host as five launches): ranges_dev[] =
 * n_ranges records { const void * src; void * dst; uint64_t bytes; } (24 bytes each, device memory, bytes % 4 == 0) copied before the token is published;
 * scratch: 4096 bytes of device memory zeroed before the first use.  Replaces the host side of the greedy loop the module hides (src/models.cpp:941-1086). */
CLLM_API int cllm_op_snapshot_argmax_set(void * stream, const void * ranges_dev, int n_ranges, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host,
                                         const void * set_table_dev, int n_set, void * scratch);

/* fused single-token attention as chatllm's eager path emits it for qlen == 1 (src/layers.cpp:2541-2561, 2499-2539):
 *   MUL_MAT(K view, Q) + SCALE(1/sqrt(hd)) + DIAG_MASK_INF + SOFT_MAX + MUL_MAT(V view, P) + PERMUTE + CONT
 * q: [hd, n_head] F32 (post-RoPE); k_cache: [max_len][n_kv_head*hd] F16; v_cache: [n_kv_head*hd][max_len] F16 (transposed);
 * the number of cached positions is *pos_dev + 1 (read on the device); out: [hd * n_head] F32.  Bit-identical to the nodes. */
CLLM_API int cllm_op_attn_decode(void * stream, const float * q, const int32_t * pos_dev, int n_head, int n_kv_head, int head_dim,
                                 const void * k_cache, const void * v_cache, int64_t max_len, float * out);

/* the whole single-token attention block of KVCacheAttention (src/layers.cpp:3044-3123 save_to_cache, 957-983 rope, 2541-2561 scores,
 * 2499-2539 probs) in one call -- 1 launch up to 512 cached positions, 3 above:
 *   ROPE(q), ROPE(k) -> SET_ROWS(k_cache row pos), CPY(v -> v_cache column pos), then the node sequence of cllm_op_attn_decode.
 * qkv: the UN-rotated projections [n_head*hd | n_kv_head*hd | n_kv_head*hd] F32; pos_dev: I32 position on the device (== n_kv - 1);
 * rope_cs: cllm_op_rope_table() of pos_dev (NULL: computed in the kernel from freq_base); rope_mode 0 (pairs i,i+1) or 2 (NEOX),
 * n_dims == head_dim, no YaRN / frequency factors (ggml_compute_forward_rope_flt, ops.cpp:5589-5865); n_kv: host copy of the cached
 * length, used only to pick the kernel; wdata: cllm_attn_decode_wsize(n_kv, ..) bytes (may be NULL when that is 0).
 * Bit-identical to the unfused nodes. */
CLLM_API int    cllm_op_rope_table(void * stream, const int32_t * pos_dev, int head_dim, float freq_base, float * cs /* head_dim floats */);
CLLM_API int    cllm_attn_decode_supported(int n_head, int n_kv_head, int head_dim, int64_t max_len);
CLLM_API size_t cllm_attn_decode_wsize(int64_t n_kv, int n_head, int64_t max_len);
CLLM_API int    cllm_op_rope_kv_attn_decode(void * stream, const float * qkv, const int32_t * pos_dev, const float * rope_cs, float freq_base, int64_t n_kv,
                                            int n_head, int n_kv_head, int head_dim, int rope_mode, void * k_cache, void * v_cache, int64_t max_len,
                                            float * out, void * wdata, size_t wsize);

typedef enum cllm_unary { CLLM_UNARY_SILU = 10 /* == GGML_UNARY_OP_SILU */ } cllm_unary;
/* GGML_OP_UNARY         ggml_vec_silu_f32 (ggml-cpu/vec.cpp:396-431) */
CLLM_API int cllm_op_unary(void * stream, int op, const cllm_tensor * src, cllm_tensor * dst);
/* GGML_OP_ADD / GGML_OP_MUL with broadcast of src1 (ggml-cpu/binary-ops.cpp) */
CLLM_API int cllm_op_add(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst);
CLLM_API int cllm_op_mul(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst);
/* GGML_OP_DIV  (ggml_vec_div_f32, vec.h:104: IEEE division, src1 broadcast like ADD / MUL) */
CLLM_API int cllm_op_div(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst);
/* GGML_OP_SUM_ROWS  ggml_compute_forward_sum_rows_f32 (ops.cpp:1451-1482): dst[0, i1, i2, i3] = sum over i0, accumulated in double */
CLLM_API int cllm_op_sum_rows(void * stream, const cllm_tensor * src, cllm_tensor * dst);
/* GGML_OP_TOP_K  ggml_compute_forward_top_k_f32 (ops.cpp:8057-8094): dst I32 [k, ...] = indices of the k largest of each row, descending,
 * first two swapped; the expert selection of GenericSparseMLP::select_experts (src/layers.cpp:3817-3840) */
CLLM_API int cllm_op_top_k(void * stream, const cllm_tensor * src, cllm_tensor * dst);
/* the tail of a sparse-MoE block (GenericSparseMLP::forward + forward_with_experts, src/layers.cpp:3792-3872) in one launch:
 *   w = GET_ROWS(probs, ids); w /= SUM_ROWS(w); dst = experts[:,0,:]*w0 + experts[:,1,:]*w1 + ... (+ resid)
 * experts F32 [H, k, T], probs F32 [n_expert, T], ids I32 [k, T], resid / dst F32 [H, T].  Bit-identical to the node sequence. */
CLLM_API int cllm_op_moe_combine(void * stream, const cllm_tensor * experts, const cllm_tensor * probs, const cllm_tensor * ids, const cllm_tensor * resid,
                                 cllm_tensor * dst);
/* the down-projection MUL_MAT_ID of ONE token over TWO slots together with the block's tail (GenericSparseMLP::forward, src/layers.cpp:3840-3872; the
 * nodes MUL_MAT_ID -> GET_ROWS -> SUM_ROWS -> DIV -> MUL -> ADD of the slot views -> ADD residual) in one launch; as [K, H, E] quantized, b F32 [K, 2, 1],
 * ids I32 [2, 1], probs F32 [E, 1], resid (may be NULL) / dst F32 [H, 1]; dst may be resid.  Bit-identical to cllm_op_mul_mat_id + cllm_op_moe_combine;
 * CLLM_E_UNSUPPORTED = use those. */
CLLM_API int cllm_op_mul_mat_id_combine(void * stream, const cllm_tensor * as, const cllm_tensor * b, const cllm_tensor * ids, const cllm_tensor * probs,
                                        const cllm_tensor * resid, cllm_tensor * dst);
/* the head of a sparse-MoE block AND its experts' gate / up projections for ONE token in one launch (GenericSparseMLP::forward, src/layers.cpp:3792-3872: RMS_NORM -> MUL ->
 * MUL_MAT(gate.weight) -> SOFT_MAX -> TOP_K -> {MUL_MAT_ID(gate), MUL_MAT_ID(up)} -> SiLU -> MUL): cllm_op_moe_router + cllm_op_mul_mat_id_silu_mul without the router launch --
 * every workgroup redoes the router behind its norm prologue.  x / norm_w F32 [K], gate_w [K, E] and as_gu [K, 2F, E] (per-expert interleaved pack) of ONE quantized type,
 * probs F32 [E] and ids I32 [k] written, dst F32 [F, k].  Bit-identical to the two calls; CLLM_E_UNSUPPORTED = make them. */
CLLM_API int cllm_op_moe_router_gate_up(void * stream, const cllm_tensor * x, const cllm_tensor * norm_w, float eps, const cllm_tensor * gate_w, const cllm_tensor * as_gu,
                                        cllm_tensor * probs, cllm_tensor * ids, cllm_tensor * dst);
/* the head of a sparse-MoE block for ONE token (GenericSparseMLP::forward, src/layers.cpp:3792-3830: the post-attention RMS_NORM -> MUL, the
 * router MUL_MAT(gate.weight), SOFT_MAX, TOP_K) in one launch: xnorm F32 [K] (the experts' input; may be x itself), probs F32 [n_expert <= 64],
 * ids I32 [k].  gate_w: dense quantized [K <= 16384, n_expert].  Bit-identical to the node sequence; CLLM_E_UNSUPPORTED = use the separate ops. */
CLLM_API int cllm_op_moe_router(void * stream, const cllm_tensor * x, const cllm_tensor * norm_w, float eps, const cllm_tensor * gate_w,
                                cllm_tensor * xnorm, cllm_tensor * probs, cllm_tensor * ids);
/* fused  dst = silu(gate) * up   (BaseMLP::forward, src/layers.cpp:2475-2483: UNARY(SILU) then MUL) */
CLLM_API int cllm_op_silu_mul(void * stream, const cllm_tensor * gate, const cllm_tensor * up, cllm_tensor * dst);

/* GGML_OP_SET_ROWS      ggml_compute_forward_set_rows_f32 (ops.cpp:4892-4940): K-cache write (F32 -> F16|F32 rows at idx) */
CLLM_API int cllm_op_set_rows(void * stream, const cllm_tensor * src, const cllm_tensor * idx, cllm_tensor * dst);
/* GGML_OP_CPY / DUP / CONT  ggml_compute_forward_dup (ops.cpp:47-330,526): same #elements, F32/F16 -> F32/F16, any strides
 * (the transposed V-cache write of src/layers.cpp:3082-3093, cache shifts, permute+cont) */
CLLM_API int cllm_op_cpy(void * stream, const cllm_tensor * src, cllm_tensor * dst);
/* GGML_OP_GET_ROWS      ggml_compute_forward_get_rows (ops.cpp:4653-4700,4820): embedding gather with dequantization */
CLLM_API int cllm_op_get_rows(void * stream, const cllm_tensor * src, const cllm_tensor * idx, cllm_tensor * dst);
/* WEIGHT quantizers on the device: quantize_row_{q8_0,q4_0,q4_1,q5_0,q5_1,q4_K}_ref and the F16 conversion (ggml/src/ggml-quants.c:36-222, 622-702, 1280-1350),
 * byte-identical to the reference's from_float_ref.  Replaces the host-side loop of ggml::from_float (src/layers.cpp:358-373) that chatllm.cpp's loader runs when a
 * tensor is re-quantized on load (src/chat.cpp:1246-1279): nrows rows of k fp32 values at x (device, contiguous, 16-byte aligned) -> rows of `type` blocks at y. */
CLLM_API int cllm_op_quantize_rows(void * stream, int type, const float * x, void * y, int64_t k, int64_t nrows);
/* dequantize_row_q4_0 / q8_0 / q4_K (ggml/src/ggml-quants.c:307-325, 401-414, 1352-1373) */
CLLM_API int cllm_dequantize_row(void * stream, int type, const void * blocks, float * y, int64_t k);

/* ---- host-side graph runner for the Llama-3 / Qwen2 decoder (chatllm.cpp_amd/csrc/decoder.hip) ------------
 * mirrors HeterogeneousModel::forward + LMBlock1Forward::forward + LMFinalSteps::forward
 * (src/models.cpp:1399-1424,1736-1784; src/layers.cpp:2719-2761) as a fixed launch sequence on one
 * stream, captured into a hipGraph per (qlen) so that the per-token host cost is one graph launch
 * (SURVEY.md 8f rank 1).  Weights are raw reference-format quant blocks, row-major [out][in].        */
typedef struct cllm_llama_config {
    int32_t n_layer, hidden, n_head, n_kv_head, head_dim, ffn, vocab, max_len;
    int32_t rope_mode;       /* 0 interleaved (Llama-3), 2 NEOX (Qwen2) */
    float   rope_theta, rms_eps;
    int32_t qkv_bias;
    int32_t tp_rank, tp_size; /* tensor-parallel shard of heads / ffn columns (1 = whole model) */
    int32_t ffn_local;        /* this rank's share of ffn; 0 = ffn / tp_size.  The down projection is cut in WHOLE quant blocks of its weight type, which need not divide evenly
                               * (BASELINE cfg4: Qwen2-72B's Q8_0 down_proj has 29568 / 32 = 924 blocks -> 8 ranks get 116, 116, 116, 116, 115, 115, 115, 115: cllm_tp_split);
                               * gate / up rows are the same features */
} cllm_llama_config;
/* whole-unit split of n_units (quant blocks, heads ...) over tp_size ranks: the first n_units % tp_size ranks hold one unit more.  *first / *count: rank's range. */
CLLM_API int  cllm_tp_split(int64_t n_units, int tp_size, int tp_rank, int64_t * first, int64_t * count);

typedef struct cllm_llama cllm_llama;   /* opaque */
typedef void (*cllm_allreduce_fn)(void * user, void * stream, float * buf, int64_t n);

CLLM_API int  cllm_llama_create(const cllm_llama_config * cfg, void * stream, cllm_llama ** out);
CLLM_API void cllm_llama_destroy(cllm_llama * m);
/* name: "tok_embd" "lm_head" "out_norm" "layers.N.{attn_norm,ffn_norm,wq,wk,wv,wo,wgate,wup,wdown,bq,bk,bv}".
 * data: HOST pointer to nbytes of blocks (type = cllm_type) ; uploaded and owned by the model.              */
CLLM_API int  cllm_llama_set_weight(cllm_llama * m, const char * name, int type, const void * data, size_t nbytes);
/* same, but `data` is a DEVICE pointer that the caller keeps alive (no copy) */
CLLM_API int  cllm_llama_bind_weight(cllm_llama * m, const char * name, int type, void * dev_data, size_t nbytes);
CLLM_API int  cllm_llama_set_allreduce(cllm_llama * m, cllm_allreduce_fn fn, void * user);
/* Tensor parallelism over RCCL/xGMI (cfg.tp_size > 1; the reference only splits by layer, SplitMethod::Row is a TODO in
 * src/backend.cpp:677-778): one all-reduce(sum) of [hidden] fp32 after o_proj and after down_proj, issued on the runner's own
 * stream, hence part of the captured decode graph.  Rank 0 creates the 128-byte id, the host broadcasts it, every rank calls
 * cllm_tp_init (collective) and binds the communicator to its model.  librccl.so is dlopen'ed on first use. */
CLLM_API int  cllm_tp_unique_id(void * out128);
CLLM_API int  cllm_tp_init(const void * id128, int rank, int nranks, void ** comm_out);
CLLM_API int  cllm_tp_comm_info(void * comm, int * nranks, int * rank);   /* ncclCommCount / ncclCommUserRank: the group as RCCL reports it */
CLLM_API int  cllm_tp_destroy(void * comm);
CLLM_API int  cllm_tp_all_reduce_f32(void * comm, void * stream, float * buf, size_t n);
CLLM_API int  cllm_llama_set_tp_comm(cllm_llama * m, void * comm);
/* One-shot direct-write all-reduce for the decode-sized messages (tp_oneshot.hip): every rank writes its partial vector as 8-byte {value, sequence number} granules
 * (16-byte write-through stores) into a slot of every peer's receive buffer (peer memory mapped through HIP IPC, one process per GPU), polls the granules of all ranks
 * in its own buffer and sums them in rank order: one kernel launch per all-reduce, no flag and no fence, inside the captured decode graph.
 * create -> exchange the 64-byte handles (rank-ordered) by any host-side means -> connect -> bind. */
CLLM_API int  cllm_tp_oneshot_create(int rank, int nranks, size_t max_n, void ** out, void * handle64);
CLLM_API int  cllm_tp_oneshot_connect(void * os, const void * handles);
CLLM_API int  cllm_tp_oneshot_all_reduce_f32(void * os, void * stream, float * buf, size_t n);
CLLM_API int  cllm_tp_oneshot_error(void * os);           /* 1: a flag wait timed out since creation (the runner checks it after every step's synchronize and fails the step) */
CLLM_API int  cllm_tp_oneshot_fine_grained(void * os);    /* 1: fine-grained (cross-GPU coherent) receive buffer; 0: coarse-grained, only accepted with CLLM_TP_ONESHOT_SAME_DEVICE=1 */
CLLM_API int  cllm_tp_oneshot_clear_error(void * os);     /* after a reported time-out, every rank at a common boundary: clears the sticky error word */
CLLM_API int  cllm_tp_oneshot_destroy(void * os);
CLLM_API int  cllm_llama_set_tp_oneshot(cllm_llama * m, void * os);
/* The all-reduce FUSED into the neighbouring mat-vecs of a single-token step (gemv_tp.hip; nothing in the reference to replace: SplitMethod::Row is a TODO,
 * src/backend.h:322-327): the o / down projections send their partial rows as granules into every rank's receive buffer, the next RMS_NORM mat-vec gathers and adds them in
 * rank order -- NO all-reduce launch (a tensor-parallel layer is 5 launches like a single-GPU one).  Receive buffers: n_sites (>= 2 n_layer) x nranks x max_n (>= hidden)
 * granules per rank; create -> exchange the handles -> connect -> cllm_llama_set_tp_fused.  Prompts (multi-token graphs) keep using the communicator / one-shot / callback. */
CLLM_API int          cllm_tp_fused_create(int rank, int nranks, int n_sites, size_t max_n, void ** out, void * handle64);
CLLM_API int          cllm_tp_fused_connect(void * os, const void * handles);
CLLM_API const void * cllm_tp_fused_dev(void * os);           /* the device-side context the kernels read */
CLLM_API int          cllm_tp_fused_sites(void * os);
CLLM_API size_t       cllm_tp_fused_max_n(void * os);
CLLM_API int          cllm_tp_fused_fine_grained(void * os);  /* as cllm_tp_oneshot_fine_grained */
CLLM_API int          cllm_tp_fused_advance(void * os, void * stream);   /* next step number: once per decode step on every rank (the runner does it) */
CLLM_API int          cllm_tp_fused_error(void * os);         /* 1: a granule wait timed out since creation */
CLLM_API int          cllm_tp_fused_destroy(void * os);
CLLM_API int          cllm_llama_set_tp_fused(cllm_llama * m, void * os);
/* The same buffers for ranks that live in ONE process (the logical tensor-parallel device of the ggml module, chatllm.cpp_amd/host/ggml-hip.cpp -- the reference's host is a
 * single process that drives every GPU, src/backend.cpp:677-778; its own slot for this is SplitMethod::Row, "TODO: WIP", src/backend.h:322-327): out[r] = rank r's object on
 * devices[r]; peers are plain pointers with peer access enabled between distinct GPUs.  Ranks may share a GPU (virtual ranks); their launches must then go to ONE stream in
 * site order.  Every object is destroyed with cllm_tp_fused_destroy. */
CLLM_API int          cllm_tp_fused_create_group(int nranks, const int * devices, int n_sites, size_t max_n, void ** out);
CLLM_API int          cllm_tp_fused_clear_error(void * os);   /* after a reported time-out: clears the sticky error word (the ranks must restart from a common step boundary) */
/* The two tensor-parallel forms of the single-column mat-vec as operators (gemv_tp.hip; what a sharded Linear::forward src/layers.cpp:2111-2129 + the residual ADD become):
 *   scatter: src0 = this rank's K-shard of an o / down projection [K_r, N]; its partial rows go out as granules of `site` into every rank's buffer (no dst).
 *            pro 2: act = quantize(px); pro 3: act = quantize(silu(px[2i]) * px[2i+1])
 *   gather:  x' = px + sum over ranks (rank order) of the granules of `site`; dst = src0 . quantize(RMS_NORM(x', eps) * pw) (+ bias | epi 1: SiLU(gate)*up over alternating
 *            rows); xout (!= px) = x' (the new residual stream, every rank computes the same bits). */
CLLM_API int          cllm_op_mul_mat_vec_tp_scatter(void * stream, const cllm_tensor * src0, int pro, const float * px, void * tp_fused, int site);
CLLM_API int          cllm_op_mul_mat_vec_tp_gather(void * stream, const cllm_tensor * src0, const float * px, const float * pw, float eps, int epi, const float * bias, float * dst,
                                                    void * tp_fused, int site, float * xout);
/* xout[0..n) = px + the all-reduced partials of `site`: the gather's residual fold alone (a step whose head is not a fusable mat-vec -- LMFinalSteps keeps the normalised hidden
 * state as a graph output, src/models.cpp:1754-1755) */
CLLM_API int          cllm_op_tp_gather_residual(void * stream, const float * px, int64_t n, void * tp_fused, int site, float * xout);
/* KV-cache shards of the logical tensor-parallel device (tp_kv.hip; the host's cache: KVCacheAttention src/layers.cpp:3044-3123): rows [p0, p1) -- or, pos_dev != NULL, the one
 * row at *pos_dev -- of columns [kd_offset, kd_offset + kd_shard) between the host's caches (K [n][kd_full], V [kd_full][max_len], F16) and a rank's dense shards
 * (K [n][kd_shard], V [kd_shard][max_len]).  table_dev: n_layers x 4 device pointers { host K, host V, shard K, shard V }. */
CLLM_API int          cllm_op_kv_shard_copy(void * stream, const void * table_dev, int n_layers, int kd_shard, int kd_full, int kd_offset, int64_t max_len, int64_t p0, int64_t p1,
                                            const int32_t * pos_dev, int to_authoritative);
/* pitched device-to-device copy: the K-split of a quantized matrix (whole quant blocks of every row) for the shards above */
CLLM_API int          cllm_copy_2d(void * stream, void * dst, size_t dst_pitch, const void * src, size_t src_pitch, size_t width_bytes, size_t rows);
/* run qlen tokens (host int32) at positions n_past..; writes logits[vocab] of the last token to
 * logits_dev (device, may be NULL) and/or logits_host (may be NULL; implies a stream sync).               */
CLLM_API int  cllm_llama_forward(cllm_llama * m, const int32_t * tokens, int qlen, int n_past, float * logits_dev,
                                 float * logits_host);
/* decode-loop helpers: greedy argmax on device feeding the next step without a host round trip */
CLLM_API int  cllm_llama_decode_greedy(cllm_llama * m, int32_t first_token, int n_past, int n_steps, int32_t * out_tokens_host);
/* one step of the FUSED single-token path without sampling (parity surface: must equal cllm_llama_forward bit for bit) */
CLLM_API int  cllm_llama_decode_fused_logits(cllm_llama * m, int32_t token, int n_past, float * logits_host);
CLLM_API int  cllm_llama_use_graph(cllm_llama * m, int enable);
/* test hook: read an internal activation buffer of the last step ("x", "qkv", "att", "gu", "logits") */
CLLM_API int  cllm_llama_debug_read(cllm_llama * m, const char * what, float * host, int64_t n);
CLLM_API size_t cllm_llama_weight_bytes(const cllm_llama * m);

#ifdef __cplusplus
}
#endif
#endif /* CHATLLM_HIP_H */
