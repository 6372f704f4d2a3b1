#pragma once
// gemv_decode_kernel.h -- k_gemv_dec, the decode step's quantized mat-vec (Q4_K / Q4_0 / Q4_1 / Q8_0 weights): one activation column, produced inside the kernel.
//
//   prologue PRO 1: act = quantize_q8_K(RMS_NORM(px) * pw)     (LMBlock1Forward: input_layernorm / post_attention_layernorm -> Linear)
//            PRO 2: act = quantize_q8_K(px)                    (attention output -> o_proj, SiLU*up -> down_proj)
//            PRO 3: act = quantize_q8_K(silu(px[2i]) * px[2i+1])  (interleaved gate/up pairs -> down_proj, BaseMLP::forward)
//            PRO 4: act = quantize_q8_K(silu(px[i]) * pw[i])      (separate gate / up vectors: the reference's own graph, fused by the module)
//   epilogue EPI 1: W rows alternate gate_u, up_u; dst[u] = silu(W[2u].act) * (W[2u+1].act)
//            EPI 3: MUL_MAT_ID(down experts) of one token with TWO slots + the block's tail (GenericSparseMLP::forward src/layers.cpp:3840-3872): a unit is
//                   one output row, its two sub-rows are the two picked experts' rows over the two slots' activations;
//                   dst[r] = (y0[r] * w0 + y1[r] * w1) (+ resid[r]), w = probs[ids] / (probs[ids[0]] + probs[ids[1]])   (pw = probs, ids = the TOP_K output)
//            EPI 5 (MOE, PRO 1): EPI 1 over the experts picked INSIDE the launch -- every workgroup redoes the block's router behind its RMS_NORM prologue (logits = Wr . act
//                   by one wave per expert, SOFT_MAX + TOP_K by one wave: EPI 2's code), workgroup (0, 0) publishes probs / ids for the down + combine launch; the
//                   router launch of its own (6.6 us of a 75 us block) disappears (padd = router weights, xout = probs out, ids = ids OUT, px_slot_stride = experts)
//            EPI 2: the sparse-MoE router (GenericSparseMLP::forward src/layers.cpp:3792-3830): ONE workgroup; the rows are the experts' logits,
//                   dst = SOFT_MAX(logits), ids (written) = TOP_K(dst, k = dst_slot_stride), xout = the normalised activation (the experts' input)
//   tensor parallel, all-reduce FUSED into the neighbouring mat-vecs (gemv_tp.hip; the reference has no tensor parallelism: SplitMethod::Row is a TODO, src/backend.h:322-327):
//            EPI 4: the row results are this rank's PARTIAL sums of an o / down projection; every row goes out as an 8-byte {value, step number} granule into this
//                   rank's slot of EVERY rank's receive buffer (system-scope write-through stores, peer memory mapped through HIP IPC) -- no dst, no all-reduce launch
//            PRO 5: PRO 1 whose residual input is  x + sum over ranks (rank order) of the granules of the previous o / down projection, polled from this rank's own
//                   receive buffer until they carry this step's number; workgroup 0 stores the new residual stream to xout
//            (`ids` = the device-side tp_fuse_dev context, `dst_slot_stride` (EPI 4) / `px_slot_stride` (PRO 5) = the site: 2 * layer + {0: o, 1: down})
//   dst[r] = W[r] . act (+ bias[r]) (+ resid[r])               (Linear::forward src/layers.cpp:2111-2129, residual adds :2740,:2758)
// Same arithmetic as RMS_NORM -> MUL -> quantize_row_q8_K -> MUL_MAT (-> ADD) on the node-by-node path, bit for bit:
// the reductions (rms_block_sumsq_1024, quant4_q8_K, q4k_step, wave_sum) are the shared definitions.
//
// What shapes this kernel (measured with the in-kernel stamps of tools/gemv_phase_probe.py):
//   * a launch starts with a cold instruction cache and its instruction fetches queue behind its own weight stream, so
//     everything before the main loop is kept short: the prologue is a template parameter (no code for the others), nothing
//     is divided at run time (the dealing of rows to waves is precomputed on the host), the arguments are individual
//     kernel parameters in the order they are needed;
//   * the activation loads are issued first, then two steps of weight prefetch, then the prologue computes while they fly
//     (a deeper burst only delays the prologue: 16 waves x 16 loads take 1.7 us just to issue);
//   * one 1024-thread workgroup per CU: 16 waves share one prologue; steady state keeps two steps per wave in flight and
//     streams at ~6.2 TB/s.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"
#include "q32.h"


// The kernel parameters are individual scalars in the order the kernel needs them: the first 16 dwords are preloaded into
// SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count=16), so the activation loads do not wait for a kernarg fetch.
//   units = rows (or gate/up row pairs), dealt as kfull full rounds of nwaves units + nrem (host-computed: no division here)


// the weight stream's cache policy: default.  (Non-temporal global_load into VGPRs measured 580 vs 681 tok/s, gate/up 18.4 vs 14.6 us -- the guide's nt gain is for the LDS-DMA
// stream, global_load_lds ... nt, not for this register path: profiles/r03_ldsdma_nt_decode_matvecs.txt.  The A/B macro is retired.)
#define TS(k) do { if (ts && threadIdx.x == 0) ts[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

// device-side context of the fused tensor-parallel all-reduce (tp_oneshot.hip cllm_tp_fused_*): receive buffers [site][rank][max_n] x 8-byte granules {value, step}
struct tp_fuse_dev { char * peer[16]; int rank, nranks; unsigned max_n, pad; const unsigned * step; unsigned * err; };
__device__ __forceinline__ u32x4 tpf_load16(const void * p) { u32x4 v; asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void tpf_issue16(u32x4 & v, const void * p) { asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=&v"(v) : "v"(p) : "memory"); }      // no wait: tpf_land8
__device__ __forceinline__ void tpf_land8(u32x4 (&h)[8]) {      // ONE wait for the eight loads in flight; the values are usable only behind this statement
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]) :: "memory");
}
__device__ __forceinline__ void tpf_land16(u32x4 (&h)[16]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]),
                                        "+v"(h[8]), "+v"(h[9]), "+v"(h[10]), "+v"(h[11]), "+v"(h[12]), "+v"(h[13]), "+v"(h[14]), "+v"(h[15]) :: "memory");
}
// the four elements e .. e + 3 of the all-reduced vector of `site`: every rank's granules from this rank's own buffer, summed in rank order (bounded wait).
// The loads of B ranks (4; 8 when there are more than four ranks) go out together and are waited for ONCE -- a load-wait per granule pair took 2 N dependent round trips of
// 0.7-1.4 us each; measured with all granules present (tools/tp_gather_probe.py): 1 / 2 / 4 ranks 1.32 / 1.36 / 1.48 us, 8 ranks 2.88 us in two batches of four.
// Only a rank whose granules do not carry this step's number yet is polled again, alone.
template <int B>
__device__ __forceinline__ void tpf_gather_batch(const tp_fuse_dev * cx, const char * own, int nranks, unsigned max_n, unsigned step, int site, int e, int r0, f32x4 & acc) {
    u32x4 h[2 * B];
#pragma unroll
    for (int i = 0; i < B; i++) {
        const int r = r0 + i < nranks ? r0 + i : nranks - 1;
        const char * src = own + ((size_t)(site * nranks + r) * max_n + (size_t) e) * 8;
        tpf_issue16(h[2 * i], src); tpf_issue16(h[2 * i + 1], src + 16);
    }
    if constexpr (B == 8) tpf_land16(h); else tpf_land8(h);
#pragma unroll
    for (int i = 0; i < B; i++) {
        if (r0 + i >= nranks) break;
        u32x4 h0 = h[2 * i], h1 = h[2 * i + 1];
        if (!(h0.y == step && h0.w == step && h1.y == step && h1.w == step)) {
            const char * src = own + ((size_t)(site * nranks + r0 + i) * max_n + (size_t) e) * 8;
            int spins = 0;
            for (;;) {
                h0 = tpf_load16(src); h1 = tpf_load16(src + 16);
                if (h0.y == step && h0.w == step && h1.y == step && h1.w == step) break;
                __builtin_amdgcn_s_sleep(1);
                // (bounded, and a time-out anywhere ends every later wait at its next look: a dead peer costs one time-out, not one per slot and launch)
                if ((++spins & 1023) == 0 && (spins > (1 << 21) || __hip_atomic_load(cx->err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(cx->err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break;
                }
            }
        }
        const f32x4 v = { __uint_as_float(h0.x), __uint_as_float(h0.z), __uint_as_float(h1.x), __uint_as_float(h1.z) };
        if (r0 + i == 0) acc = v; else { acc.x = acc.x + v.x; acc.y = acc.y + v.y; acc.z = acc.z + v.z; acc.w = acc.w + v.w; }
    }
}
__device__ __forceinline__ f32x4 tpf_gather4(const tp_fuse_dev * cx, const char * own, int nranks, unsigned max_n, unsigned step, int site, int e) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
    if (nranks > 4) { for (int r0 = 0; r0 < nranks; r0 += 8) tpf_gather_batch<8>(cx, own, nranks, max_n, step, site, e, r0, acc); }
    else tpf_gather_batch<4>(cx, own, nranks, max_n, step, site, e, 0, acc);
    return acc;
}

// FMT: CLLM_TYPE_Q4_K (8 lanes per 144-byte super-block, activation quantized to Q8_K) or CLLM_TYPE_Q4_0 / Q4_1 / Q8_0 (one lane per
// 18 / 20 / 34-byte block, activation quantized to Q8_0 / Q8_1).  nblk = weight blocks per row.
// MOE (MUL_MAT_ID for one token, ggml_compute_forward_mul_mat_id ggml-cpu.c:1432-1678): blockIdx.y is the slot; the slot's expert comes from
// device memory (ids[slot], the TOP_K node's output), W / px / dst move by the slot: dst[:, slot] = W[:, :, ids[slot]]^T . x[:, slot or 0]
// FREE (gemv_free32.hip only, the 32-weight block formats): the opt-in free-order tier -- q32_block_free instead of the chain records; every other instantiation is untouched
template <int FMT, int PRO, int EPI, int NPRE, bool MOE = false, bool FREE = false>
__global__ void __launch_bounds__(1024) k_gemv_dec(const float * __restrict__ px, const float * __restrict__ pw, const float * __restrict__ padd,
                                                        const char * __restrict__ W, int nblk, int kfull, int nrem, float eps,
                                                        float * __restrict__ dst, float * __restrict__ xout,
                                                        const float * __restrict__ bias, const float * resid, unsigned long long * ts,
                                                        const int32_t * __restrict__ ids, unsigned long long w_expert_bytes, int px_slot_stride, int dst_slot_stride) {
    // (EPI 2 reuses two parameters the non-MOE forms leave idle -- ids: where the TOP_K indices go, dst_slot_stride: k -- and keeps its 2 x 64 floats behind the
    //  chain records in the dynamic LDS, so that the kernel-argument block and the static LDS of every other instantiation stay what they were: a 16-byte
    //  longer argument block + 8 bytes of static LDS measured +2..4 % on all of them)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // ---- the argument slots BY FAMILY.  The physical signature is one and the same for every instantiation (its size and order are part of the tuned launch, see above); what a
    //      slot means in a family is said HERE, once, and the code below uses these names only: ----
    //   plain:           padd = all-reduced partial folded into the residual (PRO 1, NPRE 1), xout = where the new residual stream goes
    //   sparse MoE:      ids = the TOP_K output (MOE / EPI 3: read; EPI 2 / 5: WRITTEN), pw = the probabilities (EPI 3), padd = the router's weight rows (EPI 5),
    //                    px_slot_stride = activation stride per slot (MOE, EPI 3) | number of experts (EPI 5), dst_slot_stride = dst stride per slot (MOE, EPI 5) | k of TOP_K (EPI 2),
    //                    xout = the normalised activation out (EPI 2) | the probabilities out (EPI 5)
    //   tensor parallel: ids = the tp_fuse_dev context, px_slot_stride = the site gathered (PRO 5), dst_slot_stride = the site scattered to (EPI 4)
    const tp_fuse_dev * const tp_ctx    = (PRO == 5 || EPI == 4) ? (const tp_fuse_dev *) ids : nullptr;
    const int                 tp_site_in = px_slot_stride, tp_site_out = dst_slot_stride;
    const int32_t * const     moe_ids   = ids;                                             // MOE / EPI 3: the experts picked by an earlier launch
    int32_t * const           topk_out  = const_cast<int32_t *>(ids);                      // EPI 2 / EPI 5: where this launch writes its picks
    const int                 topk_k    = dst_slot_stride;                                 // EPI 2
    const int                 n_experts = px_slot_stride;                                  // EPI 5
    const char * const        router_w  = (const char *) padd;                             // EPI 5
    const float * const       moe_probs = pw;                                              // EPI 3
    float * const             probs_out = xout;                                            // EPI 5
    const int                 act_slot_stride = px_slot_stride, out_slot_stride = dst_slot_stride;      // MOE / EPI 3 / EPI 5: per-slot strides
    if constexpr (MOE && EPI != 5) {
        int e;
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(e) : "s"(moe_ids + blockIdx.y) : "memory");
        W += (unsigned long long)(unsigned) e * w_expert_bytes;
        px += (long) blockIdx.y * act_slot_stride; dst += (long) blockIdx.y * out_slot_stride;
    }
    const char * W0 = W, * W1 = W;
    float cw0 = 0.0f, cw1 = 0.0f;
    if constexpr (EPI == 3) {
        int e0, e1;
        asm volatile("s_load_dword %0, %2, 0x0\n\ts_load_dword %1, %2, 0x4\n\ts_waitcnt lgkmcnt(0)" : "=&s"(e0), "=&s"(e1) : "s"(moe_ids) : "memory");
        W0 = W + (unsigned long long)(unsigned) e0 * w_expert_bytes; W1 = W + (unsigned long long)(unsigned) e1 * w_expert_bytes;
        const float p0 = uniform_load_f32(moe_probs + e0), p1 = uniform_load_f32(moe_probs + e1);      // k_moe_combine's order: double sum from 0, IEEE divisions
        const float sum = (float)(((double) 0.0 + (double) p0) + (double) p1);
        cw0 = __fdiv_rn(p0, sum); cw1 = __fdiv_rn(p1, sum);
    }
    // steps of weight prefetch per wave: ONE 2304-byte step for Q4_K, two 64-block steps for the others (a third step at entry or mid-prologue only delays the prologue:
    // profiles/r05_ring_per_wave_prefetch.txt; the A/B macro of the depth is retired)
    constexpr int P = (FMT == CLLM_TYPE_Q4_K) ? 1 : 2, RU = (EPI == 1 || EPI == 3 || EPI == 5) ? 2 : 1;      // steps of prefetch (a Q4_K step is 16 super-blocks = 2304 B per wave, the others' 64 blocks)
    constexpr bool IS_K = FMT == CLLM_TYPE_Q4_K, IS_Q8 = FMT == CLLM_TYPE_Q8_0, IS_Q41 = FMT == CLLM_TYPE_Q4_1;
    constexpr int KIND = IS_K ? 256 : 32;                           // elements per weight block = activation quantization block
    constexpr int BS = IS_K ? 144 : q32_fmt<IS_K ? CLLM_TYPE_Q4_0 : FMT>::BS;      // bytes per weight block
    constexpr int BPS = IS_K ? 16 : 64;                             // blocks a wave consumes per step (Q4_K: 4 lanes per super-block, q4k_emit4)
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = nblk * KIND;

    // ---- (1) this thread's activation groups: unconditional (clamped) loads, issued before anything else ----
    // pro 3: px holds interleaved (gate_e, up_e) pairs -- two 16-byte loads cover this thread's four features
    constexpr bool NORM = PRO == 1 || PRO == 5;                     // the RMS_NORM prologues
    const float * gp = (NORM || PRO == 4) ? pw : PRO == 3 ? px + 4 : px;
    constexpr int vmul = PRO == 3 ? 2 : 1;
    const int e0 = tid * 4;
    // Q16 (the plain quantize_row_q8_K prologue in front of Q4_K weights, rows of more than 4096 values: down_proj): SIXTEEN values per lane, one 256-block per 16-lane
    // row, four blocks per wave (quant16_q8_K, quant_dev.h).  The prologue is redone by all 256 workgroups and is VALU-THROUGHPUT-bound inside each for long rows: down's
    // 56 blocks cost every CU 56 x ~90 wave-instructions with four values per lane, 14 x ~110 here (its prologue barrier opens at 2.5 instead of 3.4 us, the launch takes
    // 9.8 instead of 10.6 us: profiles/r05_prologue_quant16.txt).  NOT for 4096-value rows: there four waves would do serially what sixteen do side by side (o +0.2 us;
    // the RMS_NORM prologue of qkv / gate-up in this form: +1.2 us each -- measured, same file).
    constexpr bool Q16 = FMT == CLLM_TYPE_Q4_K && PRO == 2 && NPRE >= 4;      // (the A/B macro is retired: its other settings lost, same profile)
    constexpr int NSL = EPI == 3 ? 2 : 1;                        // activation rows (EPI 3: the two slots of a sparse-MoE block's down projection, quantized back to back)
    constexpr int NQ = (NPRE == 8 ? 2 : 1) * NSL;                // passes of 64 blocks
    const int nbt = NSL * nblk;
    const int qrow = lane >> 4, qp = lane & 15, qwave = tid >> 6;
    f32x4 vv[Q16 ? 1 : NPRE], gg[(PRO != 2 || EPI == 3) ? NPRE : 1];      // (EPI 3: gg = the second slot's activation)
    f32x4 w16[Q16 ? NQ : 1][4], g16[(Q16 && PRO == 1) ? NQ : 1][4];
    if constexpr (Q16) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int bi = (qwave + 16 * q) * 4 + qrow, sl = (NSL > 1 && bi >= nblk) ? 1 : 0, blk = bi - sl * nblk;
            const int o = (bi < nbt ? blk : 0) * 256 + 16 * qp + (NSL > 1 ? sl * act_slot_stride : 0);
#pragma unroll
            for (int i = 0; i < 4; i++) w16[q][i] = *(const f32x4 *)(px + o + 4 * i);
            if constexpr (PRO == 1) {
#pragma unroll
                for (int i = 0; i < 4; i++) g16[q][i] = *(const f32x4 *)(pw + o + 4 * i);
            }
        }
    } else {
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096, ec = e < K ? e : 0;
        vv[u] = *(const f32x4 *)(px + ec * vmul);
        if (PRO != 2) gg[u] = *(const f32x4 *)(gp + ec * vmul);
        if (EPI == 3) gg[u] = *(const f32x4 *)(px + act_slot_stride + ec);
    }
    }
    // tensor parallel (PRO 1, NPRE 1): the all-reduced partial of the previous mat-vec is added to the residual stream here
    // (x + padd feeds the norm; workgroup 0 stores it to xout, a different buffer than px) instead of in a launch of its own
    f32x4 pa = {0, 0, 0, 0}, pa16[(Q16 && PRO == 1 && NPRE == 1) ? 4 : 1];
    const bool add = PRO == 1 && NPRE == 1 && EPI != 5 && padd != nullptr;        // (EPI 5: padd carries the router's weights)
    if constexpr (PRO == 5) {
        if (ts && threadIdx.x == 0) ts[blockIdx.x * 8 + 6] = wall_clock64();      // (tools/tp_gather_probe.py: the gather's duration = stamp 0 - stamp 6)
        // the all-reduce of the previous o / down projection, folded in: every rank's partial rows wait (or arrive) as granules in this rank's buffer.  Polled BEFORE the
        // weight prefetch goes out (the polls wait with vmcnt(0): behind the prefetch they would drain it anyway) -- a tensor-parallel step is bound by these arrivals
        const tp_fuse_dev * cx = tp_ctx;
        const int nr = cx->nranks, rk = cx->rank; const unsigned mxn = cx->max_n, stp = *cx->step;
        const char * own = cx->peer[rk];
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            const int e = e0 + u * 4096;
            if (e < K) {
                const f32x4 g = tpf_gather4(cx, own, nr, mxn, stp, tp_site_in, e);
                vv[u].x = vv[u].x + g.x; vv[u].y = vv[u].y + g.y; vv[u].z = vv[u].z + g.z; vv[u].w = vv[u].w + g.w;
                if (blockIdx.x == 0) *(f32x4 *)(xout + e) = vv[u];
            }
        }
    }
    if constexpr (Q16 && PRO == 1 && NPRE == 1) {
        if (add) {
            const int blk = qwave * 4 + qrow, o = (blk < nblk ? blk : 0) * 256 + 16 * qp;
#pragma unroll
            for (int i = 0; i < 4; i++) pa16[i] = *(const f32x4 *)(padd + o + 4 * i);
        }
    } else if (add) pa = *(const f32x4 *)(padd + (e0 < K ? e0 : 0));
    // EPI 4: where this rank's partial rows go (lane r: rank r's receive buffer)
    char * tp_dst = nullptr; unsigned tp_step = 0;
    if constexpr (EPI == 4) {
        const tp_fuse_dev * cx = tp_ctx;
        const int nr = cx->nranks;
        tp_step = *cx->step;
        if (lane < nr) tp_dst = cx->peer[lane] + (size_t)(tp_site_out * nr + cx->rank) * cx->max_n * 8;
    }
    TS(0);

    // ---- (2) two steps of weight prefetch.  Units are dealt in rounds of nwaves: in a full round wave (b, w) takes unit
    //          round*nwaves + 16 b + w (a workgroup streams 16 consecutive rows); the last, partial round is dealt
    //          workgroup-interleaved (w * gridDim + b) so that every CU gets the same share of it. ----
    const int grp = IS_K ? lane >> 2 : lane >> 3, j = IS_K ? lane & 3 : lane & 7;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = gridDim.x * 16;
    const int lin = blockIdx.x * 16 + wave_in_wg, alt = wave_in_wg * gridDim.x + blockIdx.x;
    const int nmine = kfull + (alt < nrem ? 1 : 0);
    const int S = (nblk + BPS - 1) / BPS;                           // steps per row
    const unsigned nb01 = (unsigned) nblk * (unsigned) BS;
    auto unit_of = [&](int k) { return k * nwaves + (k < kfull ? lin : alt); };
    // per step and lane: Q4_K header (hh) + 16 quant bytes (qq); Q4_0 / Q8_0: fp16 scale (hh.x) + 16 (qq) [+ 16 (q2)] quant bytes
    u32x4 hh[P], qq[P], q2[(IS_Q8 || IS_K) ? P : 1];
    int ik = 0, isub = 0, is = 0;                                   // issue cursor: (unit ordinal, row of the unit, step of the row)
    auto issue = [&](int p) {                                       // unconditional: out-of-range steps re-read block 0 and are masked
        const int b = IS_K ? 16 * is + grp : 64 * is + lane;
        const bool ok = ik < nmine && b < nblk;
        const char * bp = W;
        if (EPI == 3) { if (ok) bp = (isub ? W1 : W0) + (unsigned long long)(unsigned) unit_of(ik) * nb01 + __umul24((unsigned) b, (unsigned) BS); }
        else if (ok) bp = W + (unsigned long long)(unsigned)(unit_of(ik) * RU + isub) * nb01 + __umul24((unsigned) b, (unsigned) BS);      // (b < 2^24: the 32-bit v_mul_lo_u32 runs at a quarter rate)
        if (IS_K) {
            hh[p] = *(const u32x4 *) bp;
            qq[p] = *(const u32x4 *)(bp + 16 + 32 * j);
            q2[(IS_Q8 || IS_K) ? p : 0] = *(const u32x4 *)(bp + 32 + 32 * j);
        } else {
            uint32_t t, odd;             // the aligned window as loaded; q32_align() at the point of use
            q32_load_raw<IS_K ? CLLM_TYPE_Q4_0 : FMT>(bp, qq[p], q2[IS_Q8 ? p : 0], t, odd);
            hh[p].x = t; hh[p].y = odd;
        }
        if (++is == S) { is = 0; if (++isub == RU) { isub = 0; ik++; } }
    };
    if constexpr (EPI != 5) {                                       // (EPI 5: the expert is not known yet -- the first requests go out behind the router, below)
#pragma unroll
        for (int p = 0; p < P; p++) issue(p);
    }
    TS(1);

    // ---- (3) the activation row: [RMS_NORM * weight | SiLU * up |] quantize -> LDS (act layout of common.h) ----
    float scale = 1.0f;
    if constexpr (Q16 && PRO == 1) {
        __shared__ double part[16];
        // the sum of squares as a tree over the values in registers (per lane in increasing index, DPP wave reduction, the wave partials pairwise): rms_scale's interval
        // test proves per row that the order cannot matter, else wave 0 redoes the sum in the reference's serial order (common.h)
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int blk = (qwave + 16 * q) * 4 + qrow;
            const bool valid = blk < nblk;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 v = w16[q][i];
                if constexpr (NPRE == 1) {
                    if (add) {
                        v.x = v.x + pa16[i].x; v.y = v.y + pa16[i].y; v.z = v.z + pa16[i].z; v.w = v.w + pa16[i].w;
                        if (blockIdx.x == 0 && valid) *(f32x4 *)(xout + blk * 256 + 16 * qp + 4 * i) = v;
                        w16[q][i] = v;
                    }
                }
                if (valid) { sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
            }
        }
        sum = wave_sum_d(sum);
        if (lane == 0) part[qwave] = sum;
        __syncthreads();
        const double tot = (((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]))) +
                           (((part[8] + part[9]) + (part[10] + part[11])) + ((part[12] + part[13]) + (part[14] + part[15])));
        scale = rms_scale(tot, K, eps, px, add ? padd : nullptr, part);
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int blk = (qwave + 16 * q) * 4 + qrow;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                f32x4 v = w16[q][i]; const f32x4 g = g16[q][i];
                v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w;
                if (EPI == 2) { if (blk < nblk) *(f32x4 *)(xout + blk * 256 + 16 * qp + 4 * i) = v; }
                w16[q][i] = v;
            }
        }
    } else {
    if (add) {
        vv[0].x = vv[0].x + pa.x; vv[0].y = vv[0].y + pa.y; vv[0].z = vv[0].z + pa.z; vv[0].w = vv[0].w + pa.w;
        if (blockIdx.x == 0 && e0 < K) *(f32x4 *)(xout + e0) = vv[0];
    }
    // NPRE == 1: ONE wave adds the 16 partial sums and derives the scale; the other 15 wait at a second barrier instead of redoing ~90 instructions each
    // (qkv 6.3 -> 6.1 us, decode +0.4 %: profiles/r05_prologue_scale_one_wave.txt; the A/B macro is retired)
    if (PRO == 1) {
        __shared__ double part[16];
        if constexpr (NPRE == 1) {
            __shared__ float scale_w;
            double sum = 0.0;
            if (e0 < K) { const f32x4 v = vv[0]; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
            sum = wave_sum_d(sum);
            if (lane == 0) part[tid >> 6] = sum;
            lds_barrier();
            if (tid < 64) {
                double tot = part[0];
#pragma unroll
                for (int w = 1; w < 16; w++) tot += part[w];
                const float m = rms_mean(tot, K);
                const double dl = tot * ((double)(2 * (int64_t) K + 16) * 0x1p-53);
                const bool amb = !(rms_mean(tot - dl, K) == rms_mean(tot + dl, K));
                if (tid == 0) scale_w = amb ? __int_as_float(0x7fc00000) : 1.0f / sqrtf(m + eps);
            }
            lds_barrier();
            scale = scale_w;
            if (scale != scale) {                                       // the rare ambiguous row (uniform): the reference's serial sum by wave 0 (rms_scale's fallback)
                __syncthreads();
                if (tid < 64) { const double ss = rms_serial_sumsq<false>(px, add ? padd : nullptr, K); if (tid == 0) part[0] = ss; }
                __syncthreads();
                scale = 1.0f / sqrtf(rms_mean(part[0], K) + eps);
            }
        } else {
        const double sum = NPRE == 1 ? rms_block_sumsq_1024_one(vv[0], e0 < K, part) : rms_block_sumsq_1024(px, K, vv[0], part);
        scale = rms_scale(sum, K, eps, px, add ? padd : nullptr, part);
        }
    }
    if constexpr (PRO == 5) {
        // rms_block_sumsq_1024's order over the values in registers (x is not in memory: it is px + the gathered partials); rms_scale's interval test, and for the rare
        // ambiguous row the reference's serial sum over a copy of the row in LDS (behind the chain records)
        __shared__ double part[16];
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            if (e0 + u * 4096 < K) { const f32x4 v = vv[u]; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        }
        sum = wave_sum_d(sum);
        if ((tid & 63) == 0) part[tid >> 6] = sum;
        __syncthreads();
        double tot = part[0];
#pragma unroll
        for (int w = 1; w < 16; w++) tot += part[w];
        float m = rms_mean(tot, K);
        const double dl = tot * ((double)(2 * (int64_t) K + 16) * 0x1p-53);
        if (!(rms_mean(tot - dl, K) == rms_mean(tot + dl, K))) {
            float * xs = (float *)(lds + act_row_bytes(K, KIND) + 16 * (IS_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES));
#pragma unroll
            for (int u = 0; u < NPRE; u++) { const int e = e0 + u * 4096; if (e < K) *(f32x4 *)(xs + e) = vv[u]; }
            __syncthreads();
            if (tid < 64) { const double ss = rms_serial_sumsq<false>(xs, nullptr, K); if (tid == 0) part[0] = ss; }
            __syncthreads();
            m = rms_mean(part[0], K);
        }
        scale = 1.0f / sqrtf(m + eps);
    }
    }
    const int nv = K & ~7;                                          // ggml_vec_silu_f32: polynomial body below nv, libm tail
    if constexpr (Q16) {
#pragma unroll
        for (int q = 0; q < NQ; q++) {
            const int bi = (qwave + 16 * q) * 4 + qrow, sl = (NSL > 1 && bi >= nblk) ? 1 : 0, blk = bi - sl * nblk;
            if ((qwave + 16 * q) * 4 < nbt) {                       // (wave-uniform: the tie path of quant16_q8_K holds a ballot)
                float d; int ssum;
                const u32x4 qv = quant16_q8_K(w16[q], qp, &d, &ssum);
                if (bi < nbt) {
                    char * arow = lds + (NSL > 1 ? sl * (int) act_row_bytes(K, 256) : 0);
                    *(u32x4 *)(arow + blk * 256 + 16 * qp) = qv;
                    if ((qp & 1) == 0) ((int32_t *)(arow + act_off_s(K, 256)))[blk * 8 + (qp >> 1)] = ssum;
                    if (qp == 0) ((float *)(arow + act_off_d(K)))[blk] = d;
                }
            }
        }
    } else {
#pragma unroll
    for (int u = 0; u < NPRE; u++) {                                // K % KIND == 0: whole quantization lane groups stay together
        const int e = e0 + u * 4096;
        if (e < K) {
            f32x4 v = vv[u];
            if (PRO == 3) {
                const f32x4 p0 = vv[u], p1 = gg[u];                 // (g0, u0, g1, u1), (g2, u2, g3, u3)
                v.x = silu_any(p0.x, e + 0 < nv) * p0.y; v.y = silu_any(p0.z, e + 1 < nv) * p0.w;
                v.z = silu_any(p1.x, e + 2 < nv) * p1.y; v.w = silu_any(p1.z, e + 3 < nv) * p1.w;
            }
            if (PRO == 4) {
                const f32x4 g = gg[u];
                v.x = silu_any(v.x, e + 0 < nv) * g.x; v.y = silu_any(v.y, e + 1 < nv) * g.y; v.z = silu_any(v.z, e + 2 < nv) * g.z; v.w = silu_any(v.w, e + 3 < nv) * g.w;
            }
            if (NORM) { const f32x4 g = gg[u]; v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w; }
            if (EPI == 2) *(f32x4 *)(xout + e) = v;
            quant4_store<KIND, IS_Q41>(lds, K, e, lane, v);
            if (EPI == 3) quant4_store<KIND, IS_Q41>(lds + act_row_bytes(K, KIND), K, e, lane, gg[u]);
        }
    }
    }
    TS(2);
    __syncthreads();
    TS(3);

    // ---- (4) stream the rows.  Every step turns its blocks into chain records (exact integer sums + scales); the fp32 chains of the
    //          reference's AVX2 order run over them in lanes 0..11 (q4k.h / q32.h) ----
    const q4k_sel4 L = q4k_lane_sel4(lane);
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, KIND);      // (the Q8_1 flavour has the Q8_0 geometry)
    constexpr int CHB = IS_K ? Q4K_CHAIN_BYTES : Q32_CHAIN_BYTES;
    const int arb = (int) act_row_bytes(K, KIND);
    char * chain = lds + (EPI == 3 ? 2 : 1) * arb + wave_in_wg * CHB;
    const int l16 = lane & 15;
    float acc = 0.0f, gate = 0.0f;
    if constexpr (EPI == 5) {
        // ---- the block's router (GenericSparseMLP::forward src/layers.cpp:3792-3830: MUL_MAT(gate) -> SOFT_MAX -> TOP_K), redone by every workgroup: one wave per expert row
        //      over the activation row just built, the records and chains of the row loop below; then one wave runs k_soft_max's partition and k_top_k's picks ----
        const int ne = n_experts;
        const char * Wr = router_w;
        float * r_logit = (float *)(lds + arb + 16 * CHB), * r_prob = r_logit + 64;
        int32_t * r_ids = (int32_t *)(r_prob + 64);
        for (int row = wave_in_wg; row < ne; row += 16) {
            float racc = 0.0f;
            for (int sx = 0; sx < S; sx++) {
                const int b = IS_K ? 16 * sx + grp : 64 * sx + lane;
                const bool ok = b < nblk;
                const char * bp = Wr + (unsigned long long)(unsigned) row * nb01 + __umul24((unsigned)(ok ? b : 0), (unsigned) BS);
                if (IS_K) {
                    const u32x4 h = *(const u32x4 *) bp, qa = *(const u32x4 *)(bp + 16 + 32 * j), qb = *(const u32x4 *)(bp + 32 + 32 * j);
                    q4k_emit4(h, qa, qb, lds, off_d, off_s, ok ? b : 0, ok, L, chain);
                } else {
                    u32x4 ra, rb = {0, 0, 0, 0}; uint32_t t, odd, h; u32x4 w0, w1 = {0, 0, 0, 0};
                    q32_load_raw<IS_K ? CLLM_TYPE_Q4_0 : FMT>(bp, ra, rb, t, odd);
                    q32_align<IS_K ? CLLM_TYPE_Q4_0 : FMT>(ra, rb, t, odd, h, w0, w1);
                    q32_emit<IS_K ? CLLM_TYPE_Q4_0 : FMT>(h, w0, w1, lds, off_d, off_s, ok ? b : 0, ok, lane, chain);
                }
                wave_lds_fence();
                if (IS_K) q4k_chain(chain, 8, l16, racc); else q32_chain<IS_K ? CLLM_TYPE_Q4_0 : FMT>(chain, l16, racc);
                wave_lds_fence();
            }
            const float v = chain_finish<IS_K ? 1 : IS_Q41 ? 2 : 0>(racc);
            if (lane == 0) r_logit[row] = v;
        }
        __syncthreads();
        if (wave_in_wg == 0) {
            wave_soft_max_plain(r_logit, r_prob, ne, lane);
            if (lane == 0) top_k_row(r_prob, ne, (int) gridDim.y, r_ids);
            if (blockIdx.x == 0 && blockIdx.y == 0) {                   // one workgroup publishes for the down + combine launch
                for (int i = lane; i < ne; i += 64) probs_out[i] = r_prob[i];
                if (lane == 0) for (int i = 0; i < (int) gridDim.y; i++) topk_out[i] = r_ids[i];
            }
        }
        __syncthreads();
        W += (unsigned long long)(unsigned) r_ids[blockIdx.y] * w_expert_bytes;
        dst += (long) blockIdx.y * out_slot_stride;
#pragma unroll
        for (int p = 0; p < P; p++) issue(p);
    }
    int ck = 0, csub = 0, cs = 0;                                   // consume cursor
    while (ck < nmine) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int b = IS_K ? 16 * cs + grp : 64 * cs + lane;
            const bool ok = ck < nmine && b < nblk;
            const char * arow = EPI == 3 ? lds + csub * arb : lds;
            if (IS_K) {
                if constexpr (FREE) acc = acc + q4k_block_free4(hh[p], qq[p], q2[(IS_Q8 || IS_K) ? p : 0], arow, off_d, off_s, ok ? b : 0, ok, L);
                else q4k_emit4(hh[p], qq[p], q2[(IS_Q8 || IS_K) ? p : 0], arow, off_d, off_s, ok ? b : 0, ok, L, chain);
            } else {
                uint32_t h; u32x4 w0, w1 = {0, 0, 0, 0};
                q32_align<IS_K ? CLLM_TYPE_Q4_0 : FMT>(qq[p], q2[IS_Q8 ? p : 0], hh[p].x, hh[p].y, h, w0, w1);
                if constexpr (FREE) acc = acc + q32_block_free<IS_K ? CLLM_TYPE_Q4_0 : FMT>(h, w0, w1, arow, off_d, off_s, ok ? b : 0, ok);
                else q32_emit<IS_K ? CLLM_TYPE_Q4_0 : FMT>(h, w0, w1, arow, off_d, off_s, ok ? b : 0, ok, lane, chain);
            }
            issue(p);
            if constexpr (!FREE) {                                  // every step: 16 super-blocks (Q4_K) / 64 blocks of records
                wave_lds_fence();
                if (IS_K) q4k_chain(chain, 8, l16, acc); else q32_chain<IS_K ? CLLM_TYPE_Q4_0 : FMT>(chain, l16, acc);
                wave_lds_fence();
            }
            if (++cs == S) {                                        // row complete: finish the chains, epilogue, store (lane 0)
                float v = FREE ? wave_sum(acc) : chain_finish<IS_K ? 1 : IS_Q41 ? 2 : 0>(acc);
                if (ck < nmine) {                                   // wave-uniform; bias / resid come through the scalar cache
                    const int cunit = unit_of(ck), crow = cunit * RU + csub;
                    if (EPI == 1 || EPI == 5) {
                        if (csub == 0) gate = v;
                        else if (lane == 0) dst[cunit] = silu_poly(gate) * v;
                    } else if (EPI == 2) {
                        if (lane == 0) ((float *)(lds + arb + 16 * CHB))[crow] = v;
                    } else if (EPI == 3) {                          // MUL by the slot's weight, ADD of the slot views, ADD of the residual: separate roundings
                        if (csub == 0) gate = v;
                        else {
                            float o = __fadd_rn(__fmul_rn(gate, cw0), __fmul_rn(v, cw1));
                            if (resid) o = __fadd_rn(o, uniform_load_f32(resid + cunit));
                            if (lane == 0) dst[cunit] = o;
                        }
                    } else if (EPI == 4) {
                        const unsigned long long gr = ((unsigned long long) tp_step << 32) | (unsigned long long) __float_as_uint(lane_f(v, 0));
                        if (tp_dst) __hip_atomic_store((unsigned long long *)(tp_dst + (size_t) crow * 8), gr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    } else {
                        if (bias)  v = v + uniform_load_f32(bias + crow);
                        if (resid) v = v + uniform_load_f32(resid + crow);
                        if (lane == 0) dst[crow] = v;
                    }
                }
                acc = 0.0f; cs = 0;
                if (++csub == RU) { csub = 0; ck++; }
            }
        }
    }
    TS(4);
    if constexpr (EPI == 2) {                                       // SOFT_MAX -> TOP_K over the logits: one wave, the partition of k_soft_max / the picks of k_top_k
        __syncthreads();
        if (wave_in_wg == 0) {
            const int n = kfull * 16 + nrem;
            float * r_logit = (float *)(lds + arb + 16 * CHB), * r_prob = r_logit + 64;
            wave_soft_max_plain(r_logit, r_prob, n, lane);
            for (int i = lane; i < n; i += 64) dst[i] = r_prob[i];
            if (lane == 0) top_k_row(r_prob, n, topk_k, topk_out);
        }
    }
    if (ts) { __syncthreads(); TS(5); }
}
#undef TS

