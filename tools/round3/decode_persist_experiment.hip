// decode_persist.hip -- ONE launch for all transformer layers of a decode step (batch 1, Q4_K weights): the five phases of a layer
//   [RMS_NORM + quantize + q|k|v mat-vec] [RoPE + cache write + attention] [quantize + o mat-vec + residual] [RMS_NORM + quantize + gate/up mat-vec + SiLU*up]
//   [quantize + down mat-vec + residual]
// run inside one persistent kernel -- one 1024-thread workgroup per CU, a device-wide barrier between phases -- instead of five launches per layer.
// The arithmetic is that of the per-phase kernels (k_gemv_dec, k_attn_dec: the same shared definitions q4k_emit4 / q4k_chain / chain_finish, quant4_q8_K,
// rms_block_sumsq_1024*, vd32_reduce ...), so the logits keep the bits of the reference's CPU run; what changes is what happens BETWEEN the phases:
//   * a launch boundary costs ~2.2 us of dispatch / ramp / drain and the next kernel starts with a cold instruction cache and an empty memory pipeline;
//     here the instruction stream stays hot over the 32 layers and the first weight steps of phase N+1 are requested BEFORE the wait for phase N's barrier:
//     weights do not depend on activations, so the HBM stream keeps running through the dependency edge (the activation vector's hand-off);
//   * hand-off of the activation vectors between workgroups on different XCDs (private, non-coherent L2s): producers store write-through (agent-scope relaxed atomic
//     stores = sc1), drain (s_waitcnt vmcnt(0)), then arrive; consumers read with agent-scope relaxed atomic loads after the barrier
//     (cdna_hip_programming.md Guideline 16, recipe R1).  No cache-wide invalidate, no fence.  Mutable data is never read through the scalar cache.
//   * the barrier: per-XCD-group arrival counters -> one top counter -> per-group generation words (monotonic inside a launch, zeroed by a memset node before
//     every launch), one polling lane per workgroup with s_sleep, every spin bounded: a barrier that cannot complete sets an error word and the kernel winds
//     down instead of hanging the GPU (the host checks the word).
// Scope: Llama-style blocks without biases, every projection Q4_K, head size 128, H <= 4096 (one prologue group per thread), F <= 16384, up to
// attn_long_threshold() cached positions (the exact one-workgroup-per-head attention); anything else takes the five-launch path (decoder.hip decides).
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

#include <stdio.h>
#include <stdlib.h>

#ifndef PG_P
#define PG_P 2          // weight steps (16 super-blocks = 2304 B per wave) in flight per wave; they are issued BEFORE the wait on the previous phase's barrier
#endif

// ---- device-wide barrier -------------------------------------------------------------------------------------------------------------------------------------
struct pg_bar {                 // every polled word on its own 128-byte line
    unsigned cnt[2][8 * 32];    // arrivals of the workgroups with blockIdx % 8 == x (the dispatcher's XCD round-robin: contention stays inside an XCD; correctness does not
                                // depend on it), one set per phase PARITY: a workgroup without work in a phase (attention: 2 n_head workgroups) arrives for it right after
                                // arriving for the previous one, so arrivals of two consecutive phases interleave -- never of three: the arrival for phase p + 2 comes after
                                // the wait for p + 1, which every workgroup reaches only after phase p completed
    unsigned top[2][32];        // groups that completed the phase, per parity
    unsigned gen[8 * 32];       // last completed phase (atomic max), one copy per group
    unsigned err[32];           // != 0: a wait timed out (phase number) -- every later wait returns at once
};
#define PG_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ void pg_arrive(pg_bar * b, unsigned phase) {                     // ONE thread, after the workgroup's stores are drained; phase >= 1
    const unsigned x = blockIdx.x & 7, nx = (gridDim.x - x + 7) >> 3, s = phase & 1, k = (phase + 1) >> 1;      // k: phases of this parity up to and including this one
    const unsigned old = __hip_atomic_fetch_add(&b->cnt[s][x * 32], 1u, PG_RLX);
    if (old + 1 == k * nx) {
        const unsigned ngrp = gridDim.x < 8 ? gridDim.x : 8;
        const unsigned o2 = __hip_atomic_fetch_add(&b->top[s][0], 1u, PG_RLX);
        if (o2 + 1 == k * ngrp) for (unsigned i = 0; i < ngrp; i++) __hip_atomic_fetch_max(&b->gen[i * 32], phase, PG_RLX);      // (max: the releases of two consecutive phases come from different threads)
    }
}
__device__ __forceinline__ bool pg_wait(pg_bar * b, unsigned phase) {                       // ONE thread
    const unsigned x = blockIdx.x & 7;
    for (unsigned spins = 0; spins < (1u << 20); spins++) {                                  // ~1 s at worst
        if (__hip_atomic_load(&b->gen[x * 32], PG_RLX) >= phase) return true;
        if ((spins & 1023) == 1023 && __hip_atomic_load(&b->err[0], PG_RLX) != 0) return false;
#ifndef PG_NOSLEEP
        __builtin_amdgcn_s_sleep(2);
#endif
    }
    __hip_atomic_store(&b->err[0], phase, PG_RLX);
    return false;
}
// workgroup-level: everything this workgroup stored (write-through) is performed, then one lane arrives
__device__ __forceinline__ void pg_publish(pg_bar * b, unsigned phase) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) pg_arrive(b, phase);
}
// returns false (workgroup-uniform) if the barrier failed; contains a __syncthreads()
__device__ __forceinline__ bool pg_acquire(pg_bar * b, unsigned phase, int * flag_lds) {
    if (threadIdx.x == 0) *flag_lds = pg_wait(b, phase) ? 1 : 0;
    __syncthreads();
    const bool ok = *(volatile int *) flag_lds != 0;
    return ok;
}
// ---- coherent (write-through / L1-bypassing) accesses to the activation vectors ----
__device__ __forceinline__ void st_coh(float * p, float v) { __hip_atomic_store((unsigned *) p, __float_as_uint(v), PG_RLX); }
__device__ __forceinline__ float ld_coh(const float * p) { return __uint_as_float(__hip_atomic_load((const unsigned *) p, PG_RLX)); }
__device__ __forceinline__ f32x4 ld_coh4(const float * p) {                                 // p 16-byte aligned
    const unsigned long long a = __hip_atomic_load((const unsigned long long *) p, PG_RLX), b = __hip_atomic_load((const unsigned long long *) p + 1, PG_RLX);
    return f32x4{ __uint_as_float((unsigned) a), __uint_as_float((unsigned)(a >> 32)), __uint_as_float((unsigned) b), __uint_as_float((unsigned)(b >> 32)) };
}

__device__ __forceinline__ float pg_silu(float x) { return x / (1.0f + ggml_expf_poly(0.0f - x)); }

struct pg_op { int nblk, kfull, nrem; };       // K / 256; units (rows or gate/up row pairs) dealt as kfull full rounds of nwaves + nrem (host-computed)
struct pg_layer {
    const char * wqkv, * wo, * wgu, * wdown;   // Q4_K: q|k|v rows packed, o, gate/up rows interleaved, down
    const float * attn_norm, * ffn_norm;
    uint16_t * k_cache, * v_cache;
};
struct pg_args {
    const pg_layer * layers; int n_layer;
    float * x, * qkv, * att, * g;              // residual stream [H], projections [QD + 2 KD], attention output [QD], SiLU(gate)*up [F]
    const int32_t * pos_dev; const float * rope_cs;
    int nh, nkv, ML; float eps, scale;
    pg_op qkv_op, o_op, gu_op, down_op;
    pg_bar * bar;
    unsigned long long * ts;                   // optional: wall-clock stamps of workgroup 0 (tools)
};

// ---- one mat-vec phase: k_gemv_dec's Q4_K form (gemv_decode_kernel.h) with the weight prefetch ahead of the barrier wait ----------------------------------
//   PRO 1: act = quantize_q8_K(RMS_NORM(px) * pw);  PRO 2: act = quantize_q8_K(px)
//   EPI 0: dst[r] = W[r] . act (+ resid[r]);  EPI 1: rows alternate gate_u, up_u: dst[u] = silu(W[2u] . act) * (W[2u+1] . act)
template <int PRO, int EPI, int NPRE, int P>
__device__ __forceinline__ bool pg_gemv(char * lds, double * part, int * flag, const float * px, const float * __restrict__ pw, const char * __restrict__ W, const pg_op op, float eps,
                                        float * dst, const float * resid, pg_bar * bar, unsigned wait_phase, unsigned long long * tsp) {
#define PG_TSP(k) do { if (tsp && blockIdx.x == 0 && threadIdx.x == 0) tsp[k] = wall_clock64(); } while (0)
    constexpr int RU = EPI == 1 ? 2 : 1;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));             // opaque per phase: the lane constants derived from it must not be hoisted out of the layer loop (five phases' worth of them
                                               // live across the whole kernel cost 60+ spilled registers)
    const int tid = tid_, lane = tid & 63;
    const int nblk = op.nblk, kfull = op.kfull, nrem = op.nrem;
    const int K = nblk * 256;
    // ---- (1) weight prefetch: does not depend on the previous phase ----
    const int grp = lane >> 2, j = lane & 3;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = gridDim.x * 16;
    const int lin = blockIdx.x * 16 + wave_in_wg, alt = wave_in_wg * gridDim.x + blockIdx.x;
    const int nmine = kfull + (alt < nrem ? 1 : 0);
    const int S = (nblk + 15) / 16;
    const unsigned nb01 = (unsigned) nblk * 144u;
    auto unit_of = [&](int k) { return k * nwaves + (k < kfull ? lin : alt); };
    u32x4 hh[P], qq[P], q2[P];
    int ik = 0, isub = 0, is = 0;
    auto issue = [&](int p) {
        const int b = 16 * is + grp;
        const bool ok = ik < nmine && b < nblk;
        const char * bp = W;
        if (ok) bp = W + (unsigned long long)(unsigned)(unit_of(ik) * RU + isub) * nb01 + __umul24((unsigned) b, 144u);
        hh[p] = *(const u32x4 *) bp;
        qq[p] = *(const u32x4 *)(bp + 16 + 32 * j);
        q2[p] = *(const u32x4 *)(bp + 32 + 32 * j);
        if (++is == S) { is = 0; if (++isub == RU) { isub = 0; ik++; } }
    };
#pragma unroll
    for (int p = 0; p < P; p++) issue(p);

    // ---- (2) the previous phase's outputs are complete ----
    if (!pg_acquire(bar, wait_phase, flag)) return false;
    PG_TSP(2);

    // ---- (3) the activation row: [RMS_NORM * weight |] quantize -> LDS ----
    const int e0 = tid * 4;
    if constexpr (PRO == 2) {                   // plain quantization: every 256-block is independent -- group by group, nothing held in registers
#pragma unroll 1
        for (int u = 0; u < NPRE; u++) {
            const int e = e0 + u * 4096;
            if (e < K) quant4_store<256, false>(lds, K, e, lane, ld_coh4(px + e));       // (K % 256 == 0: whole waves drop out together)
        }
    } else {
    f32x4 vv[NPRE], gg[PRO == 1 ? NPRE : 1];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096, ec = e < K ? e : 0;
        vv[u] = ld_coh4(px + ec);
        if (PRO == 1) gg[u] = *(const f32x4 *)(pw + ec);
    }
    float scale = 1.0f;
    if (PRO == 1) {
        double sum = 0.0;
        if (NPRE == 1) sum = rms_block_sumsq_1024_one(vv[0], e0 < K, part);
        else {                                  // rms_block_sumsq_1024's order: thread t owns groups t, t + 1024, ... in increasing index
#pragma unroll
            for (int u = 0; u < NPRE; u++) if (e0 + u * 4096 < K) { const f32x4 v = vv[u]; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
            sum = wave_sum_d(sum);
            if ((tid & 63) == 0) part[tid >> 6] = sum;
            __syncthreads();
            double tot = part[0];
#pragma unroll
            for (int w = 1; w < 16; w++) tot += part[w];
            sum = tot;
        }
        scale = rms_scale<true>(sum, K, eps, px, nullptr, part);
    }
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096;
        if (e < K) {
            f32x4 v = vv[u];
            if (PRO == 1) { const f32x4 g = gg[u]; v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w; }
            quant4_store<256, false>(lds, K, e, lane, v);
        }
    }
    }
    __syncthreads();
    PG_TSP(3);

    // ---- (4) stream the rows ----
    const q4k_sel4 L = q4k_lane_sel4(lane);
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 256);
    const int arb = (int) act_row_bytes(K, 256);
    char * chain = lds + arb + wave_in_wg * Q4K_CHAIN_BYTES;
    const int l16 = lane & 15;
    float acc = 0.0f, gate = 0.0f, rv = 0.0f;
    int ck = 0, csub = 0, cs = 0;
    while (ck < nmine) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int b = 16 * cs + grp;
            const bool ok = ck < nmine && b < nblk;
            // the row's residual element: requested when the row starts (a coherent vector load -- the scalar cache is not coherent with this launch's own
            // stores), older than the weight steps the row still waits for, so it has landed by the row's end
            if (EPI == 0 && resid && cs == 0 && ck < nmine) rv = ld_coh(resid + unit_of(ck));
            q4k_emit4(hh[p], qq[p], q2[p], lds, off_d, off_s, ok ? b : 0, ok, L, chain);
            issue(p);
            wave_lds_fence();
            q4k_chain(chain, 8, l16, acc);
            wave_lds_fence();
            if (++cs == S) {
                const float v = chain_finish<1>(acc);
                if (ck < nmine) {
                    const int cunit = unit_of(ck), crow = cunit * RU + csub;
                    if (EPI == 1) {
                        if (csub == 0) gate = v;
                        else if (lane == 0) st_coh(dst + cunit, pg_silu(gate) * v);
                    } else {
                        float o = v;
                        if (resid) o = o + rv;
                        if (lane == 0) st_coh(dst + crow, o);
                    }
                }
                acc = 0.0f; cs = 0;
                if (++csub == RU) { csub = 0; ck++; }
            }
        }
    }
    PG_TSP(4);
#undef PG_TSP
    return true;
}

// ---- the attention phase: k_attn_dec<128, MODE, 2> (decode_fused.hip) for workgroup (hx, g, part) ------------------------------------------------------------
__device__ __forceinline__ float pg_lane_xor4_f(float v) { return __int_as_float(lane_xor4_i(__float_as_int(v))); }
__device__ __forceinline__ float pg_vd32_reduce(float a0, float a1) {
    a0 = a0 + dpp_f<DPP_ROW_ROR8>(a0); a1 = a1 + dpp_f<DPP_ROW_ROR8>(a1);
    a0 = a0 + pg_lane_xor4_f(a0);      a1 = a1 + pg_lane_xor4_f(a1);
    a0 = a0 + dpp_f<DPP_QUAD_XOR2>(a0); a1 = a1 + dpp_f<DPP_QUAD_XOR2>(a1);
    const float u = a0 + a1;
    return u + dpp_f<DPP_QUAD_XOR1>(u);
}
template <int MODE>
__device__ __forceinline__ bool pg_attn(float * sm, double * red_d, float * red_f, int * flag, const float * qkv, int pos, const float * __restrict__ rope_cs, int nh, int nkv, float scale,
                                        uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache, int ML, float * att, int hx, int g, int part, pg_bar * bar, unsigned wait_phase) {
    constexpr int HD = 128, half = HD / 2, off = MODE == 0 ? 1 : half, U = 4, PARTS = 2;
    const int r2 = nh / nkv, h = g * r2 + hx;
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));             // (see pg_gemv)
    const int tid = tid_, lane = tid & 63, wave = tid >> 6;
    const int KD = nkv * HD, QD = nh * HD;
    float * qs = sm; float * knew = sm + HD; float * vnew = sm + 2 * HD; float * sc = sm + 3 * HD;
    const int n_kv = pos + 1;
    // ---- first batch of cache rows: they depend on the position only (issued before the barrier wait) ----
    constexpr int NCH = HD / 32, VU = HD / 64 / PARTS, VPF = 8;
    const int vrow0 = part * VU * 64;
    const int c16 = lane & 15, sub = lane >> 4;
    const int ib0 = wave * 4 + sub;
    uint32_t kr0[U][NCH];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int i0 = ib0 + u * 64;
#pragma unroll
        for (int i = 0; i < NCH; i++) kr0[u][i] = *(const uint32_t *)(k_cache + (int64_t)(i0 < pos ? i0 : 0) * KD + g * HD + 32 * i + 2 * c16);
    }
    const int np = n_kv & ~31, nch = np >> 5;
    uint32_t vc0[VU][VPF];
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const uint16_t * vr = v_cache + ((int64_t) g * HD + vrow0 + ib0 + u * 64) * ML;
#pragma unroll
        for (int i = 0; i < VPF; i++) vc0[u][i] = *(const uint32_t *)(vr + 32 * (i < nch ? i : 0) + 2 * c16);
    }
    if (!pg_acquire(bar, wait_phase, flag)) return false;

    // ---- this head's projections (coherent loads) + the cos/sin of its pairs ----
    float px0 = 0.0f, px1 = 0.0f, pc = 0.0f, ps = 0.0f;
    const bool is_pair = tid < 2 * half, is_v = !is_pair && tid < 2 * half + HD;
    const int which = tid >= half ? 1 : 0, pi = tid - which * half, ic = MODE == 0 ? 2 * pi : pi;
    if (is_pair) {
        const float * x = which == 0 ? qkv + h * HD : qkv + QD + g * HD;
        px0 = ld_coh(x + ic); px1 = ld_coh(x + ic + off);
        pc = rope_cs[2 * pi]; ps = rope_cs[2 * pi + 1];
    } else if (is_v) px0 = ld_coh(qkv + QD + KD + g * HD + (tid - 2 * half));
    // ---- RoPE, fp16 rounding, cache write ----
    if (is_pair) {
        const float y0 = rope_rot_a(px0, px1, pc, ps), y1 = rope_rot_b(px0, px1, pc, ps);
        float * o = which == 0 ? qs : knew;
        o[ic] = h2f(f2h(y0)); o[ic + off] = h2f(f2h(y1));
    } else if (is_v) vnew[tid - 2 * half] = h2f(f2h(px0));
    lds_barrier();
    if (hx == 0 && part == 0 && tid < HD) {
        k_cache[(int64_t) pos * KD + g * HD + tid] = f2h(knew[tid]);
        v_cache[((int64_t) g * HD + tid) * ML + pos] = f2h(vnew[tid]);
    }
    // ---- scores[i] = K[i] . q * scale ----
    for (int ib = ib0; ib < n_kv; ib += U * 64) {
        uint32_t r[U][NCH];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i0 = ib + u * 64;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                r[u][i] = kr0[u][i];
                if (ib != ib0) r[u][i] = *(const uint32_t *)(k_cache + (int64_t)(i0 < pos ? i0 : 0) * KD + g * HD + 32 * i + 2 * c16);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int i0 = ib + u * 64;
            if (i0 >= n_kv) continue;
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int i = 0; i < NCH; i++) {
                const int e = 32 * i + 2 * c16;
                const float k0 = i0 == pos ? knew[e] : h2f((uint16_t)(r[u][i] & 0xffff)), k1 = i0 == pos ? knew[e + 1] : h2f((uint16_t)(r[u][i] >> 16));
                a0 = __builtin_fmaf(k0, qs[e], a0); a1 = __builtin_fmaf(k1, qs[e + 1], a1);
            }
            const float v = pg_vd32_reduce(a0, a1);
            if (c16 == 0) sc[i0] = v * scale;
        }
    }
    lds_barrier();
    uint32_t vc1[VU][VPF]; uint16_t vtl[VU][2];
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const uint16_t * vr = v_cache + ((int64_t) g * HD + vrow0 + ib0 + u * 64) * ML;
#pragma unroll
        for (int i = 0; i < VPF; i++) vc1[u][i] = *(const uint32_t *)(vr + 32 * (VPF + i < nch ? VPF + i : 0) + 2 * c16);
#pragma unroll
        for (int t = 0; t < 2; t++) { const int e = np + 2 * c16 + t; vtl[u][t] = vr[e < n_kv ? e : 0]; }
    }
    // ---- soft_max (k_soft_max's partition and order) ----
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += 1024) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    lds_barrier();
    mx = red_f[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red_f[w]);
    const int nv = n_kv & ~7;
    for (int i = tid; i < n_kv; i += 1024) sc[i] = i < nv ? ggml_expf_poly(sc[i] - mx) : libm_expf(sc[i] - mx);
    lds_barrier();
    if (wave == 0) {
        double sum = 0.0;
        for (int gi = lane * 8; gi < nv; gi += 64 * 8) {
            const f32x4 lo = *(const f32x4 *)(sc + gi), up = *(const f32x4 *)(sc + gi + 4);
            const float a0 = lo.x + up.x, a1 = lo.y + up.y, a2 = lo.z + up.z, a3 = lo.w + up.w;
            sum += (double)((a0 + a2) + (a1 + a3));
        }
        if (lane == 0) for (int i = nv; i < n_kv; i++) sum += (double) sc[i];
        sum = wave_sum_d(sum);
        if (lane == 0) red_d[0] = sum;
    }
    lds_barrier();
    const float inv = (float)(1.0 / red_d[0]);
    for (int i = tid; i < n_kv; i += 1024) sc[i] = h2f(f2h(sc[i] * inv));
    lds_barrier();
    // ---- ctx = V . P ----
    float * tailp = sc + ML + (wave * 4 + sub) * 32;
    const int ntail = n_kv - np;
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const int d0 = vrow0 + ib0 + u * 64;
        const uint16_t * vr = v_cache + ((int64_t) g * HD + d0) * ML;
        const float vfresh = vnew[d0];
        float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
        for (int i = 0; i < VPF; i++) {
            if (i < nch) {
                const int e = 32 * i + 2 * c16;
                const float v0 = e == pos ? vfresh : h2f((uint16_t)(vc0[u][i] & 0xffff)), v1 = e + 1 == pos ? vfresh : h2f((uint16_t)(vc0[u][i] >> 16));
                a0 = __builtin_fmaf(v0, sc[e], a0); a1 = __builtin_fmaf(v1, sc[e + 1], a1);
            }
        }
        for (int i8 = VPF; i8 < nch; i8 += 8) {
            uint32_t rr[8];
#pragma unroll
            for (int i = 0; i < 8; i++) { rr[i] = vc1[u][i]; if (i8 != VPF) rr[i] = *(const uint32_t *)(vr + 32 * (i8 + i < nch ? i8 + i : 0) + 2 * c16); }
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (i8 + i < nch) {
                    const int e = 32 * (i8 + i) + 2 * c16;
                    const float v0 = e == pos ? vfresh : h2f((uint16_t)(rr[i] & 0xffff)), v1 = e + 1 == pos ? vfresh : h2f((uint16_t)(rr[i] >> 16));
                    a0 = __builtin_fmaf(v0, sc[e], a0); a1 = __builtin_fmaf(v1, sc[e + 1], a1);
                }
            }
        }
        const float res = pg_vd32_reduce(a0, a1);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int e = np + 2 * c16 + t;
            if (e < n_kv) tailp[2 * c16 + t] = (e == pos ? vfresh : h2f(vtl[u][t])) * sc[e];
        }
        wave_lds_fence();
        if (c16 == 0) {
            double s = (double) res;
            for (int t = 0; t < ntail; t++) s += (double) tailp[t];
            st_coh(att + h * HD + d0, (float) s);
        }
        wave_lds_fence();
    }
    return true;
}

template <int MODE, int NPRE_DOWN>
__global__ void __launch_bounds__(1024) k_decode_layers(const pg_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ double part[16];
    __shared__ double red_d[1];
    __shared__ float red_f[16];
    __shared__ int flag[1];
    pg_bar * bar = a.bar;
    unsigned phase = 0;                        // phases completed so far (the barrier's generation)
    int pos;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pos) : "s"(a.pos_dev) : "memory");      // constant during the launch
    const int r2 = a.nh / a.nkv, n_attn = a.nh * 2;
    const bool attn_wg = (int) blockIdx.x < n_attn;
    const int hx = blockIdx.x % r2, ag = (blockIdx.x / r2) % a.nkv, apart = blockIdx.x / (r2 * a.nkv);
#define PG_TS(k) do { if (a.ts && blockIdx.x == 0 && threadIdx.x == 0) a.ts[(il * 5 + ph) * 8 + (k)] = wall_clock64(); } while (0)
#define PG_TSPTR (a.ts ? a.ts + (il * 5 + ph) * 8 : nullptr)
    for (int il = 0; il < a.n_layer; il++) {
        const pg_layer & L = a.layers[il];
        int ph = 0;
        // [norm + quantize + q|k|v]   (waits for the previous layer's down projection; the first layer's x comes from the embedding launch: phase 0 is complete at once)
        PG_TS(0);
        if (!pg_gemv<1, 0, 1, PG_P>(lds, part, flag, a.x, L.attn_norm, L.wqkv, a.qkv_op, a.eps, a.qkv, nullptr, bar, phase, PG_TSPTR)) return;
        pg_publish(bar, ++phase); PG_TS(1); ph = 1;
        // [RoPE + cache write + attention]
        PG_TS(0);
        if (attn_wg) { if (!pg_attn<MODE>((float *) lds, red_d, red_f, flag, a.qkv, pos, a.rope_cs, a.nh, a.nkv, a.scale, L.k_cache, L.v_cache, a.ML, a.att, hx, ag, apart, bar, phase)) return; }
        pg_publish(bar, ++phase); PG_TS(1); ph = 2;
        // [quantize + o + residual]
        PG_TS(0);
        if (!pg_gemv<2, 0, 1, 1>(lds, part, flag, a.att, nullptr, L.wo, a.o_op, a.eps, a.x, a.x, bar, phase, PG_TSPTR)) return;
        pg_publish(bar, ++phase); PG_TS(1); ph = 3;
        // [norm + quantize + gate/up + SiLU * up]
        PG_TS(0);
        if (!pg_gemv<1, 1, 1, PG_P>(lds, part, flag, a.x, L.ffn_norm, L.wgu, a.gu_op, a.eps, a.g, nullptr, bar, phase, PG_TSPTR)) return;
        pg_publish(bar, ++phase); PG_TS(1); ph = 4;
        // [quantize + down + residual]
        PG_TS(0);
        if (!pg_gemv<2, 0, NPRE_DOWN, PG_P>(lds, part, flag, a.g, nullptr, L.wdown, a.down_op, a.eps, a.x, a.x, bar, phase, PG_TSPTR)) return;
        pg_publish(bar, ++phase); PG_TS(1);
    }
#undef PG_TS
#undef PG_TSPTR
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------------------
size_t decode_layers_state_bytes(int n_layer) { return sizeof(pg_bar) + (size_t) n_layer * sizeof(pg_layer); }
// OPT-IN (CLLM_DECODE_PERSIST=1): measured SLOWER than the five launches per layer on MI355X (616 vs 714 tok/s, profiles/r03_decode_persistent_launch.txt): a device-wide
// barrier under the weight stream costs more than a launch boundary, and weight loads requested ahead of the barrier wait delay the barrier's own loads
static int g_persist = -1;
bool decode_layers_enabled() { if (g_persist < 0) g_persist = getenv("CLLM_DECODE_PERSIST") && atoi(getenv("CLLM_DECODE_PERSIST")) != 0; return g_persist != 0; }
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_decode_persist(int on) { g_persist = on ? 1 : 0; }      // tests: both decode paths inside one process

// CLLM_E_UNSUPPORTED (nothing launched): the five-launch path takes the step.  state: decode_layers_state_bytes() of device memory owned by the caller;
// layer_tab[i] = { wqkv, wo, wgu, wdown, attn_norm, ffn_norm, k_cache, v_cache } (8 pointers per layer, host memory; uploaded once: layers_ready)
int launch_decode_layers(hipStream_t st, void * state, bool * layers_ready, const void * const * layer_tab, int n_layer, int H, int nh, int nkv, int hd, int F, int ML, int rope_mode,
                         float eps, float * x, float * qkv, float * att, float * g, const int32_t * pos_dev, const float * rope_cs, unsigned long long * ts) {
    if (!decode_layers_enabled() || hd != 128 || nh % nkv || H % 256 || F % 256 || H > 4096 || F > 16384 || (nh * hd) % 256 || nh * hd > 4096 || ML % 8 || n_layer <= 0) return CLLM_E_UNSUPPORTED;
    const int cus = device_cu_count();
    if (nh * 2 > cus) return CLLM_E_UNSUPPORTED;
    const int QD = nh * hd, KD = nkv * hd;
    const int grid = cus;
    const size_t lds_mv = act_row_bytes(F > H ? F : H, 256) + 16 * (size_t) Q4K_CHAIN_BYTES;
    const size_t lds_at = (size_t)(3 * hd + ML) * 4 + 64 * 32 * 4;
    const size_t lds = lds_mv > lds_at ? lds_mv : lds_at;
    if (lds > 150 * 1024) return CLLM_E_UNSUPPORTED;
    char * sb = (char *) state;
    pg_layer * dl = (pg_layer *)(sb + sizeof(pg_bar));
    if (!*layers_ready) {
        std::vector<pg_layer> hl(n_layer);
        for (int i = 0; i < n_layer; i++) {
            const void * const * t = layer_tab + 8 * i;
            hl[i].wqkv = (const char *) t[0]; hl[i].wo = (const char *) t[1]; hl[i].wgu = (const char *) t[2]; hl[i].wdown = (const char *) t[3];
            hl[i].attn_norm = (const float *) t[4]; hl[i].ffn_norm = (const float *) t[5]; hl[i].k_cache = (uint16_t *) t[6]; hl[i].v_cache = (uint16_t *) t[7];
        }
        HIP_TRY(hipMemcpy(dl, hl.data(), (size_t) n_layer * sizeof(pg_layer), hipMemcpyHostToDevice));
        *layers_ready = true;
    }
    auto mkop = [&](int K, int units) { pg_op o; o.nblk = K / 256; const int nwaves = grid * 16; o.kfull = units / nwaves; o.nrem = units % nwaves; return o; };
    pg_args a;
    a.layers = dl; a.n_layer = n_layer; a.x = x; a.qkv = qkv; a.att = att; a.g = g; a.pos_dev = pos_dev; a.rope_cs = rope_cs;
    a.nh = nh; a.nkv = nkv; a.ML = ML; a.eps = eps; a.scale = 1.0f / sqrtf((float) hd);
    a.qkv_op = mkop(H, QD + 2 * KD); a.o_op = mkop(QD, H); a.gu_op = mkop(H, F); a.down_op = mkop(F, H);
    a.bar = (pg_bar *) sb; a.ts = ts;
    HIP_TRY(hipMemsetAsync(sb, 0, sizeof(pg_bar), st));                 // every polled word starts at zero (a memset node: replayed with the graph)
    const bool npre4 = F > 4096;
#define GOP(MODE_, NP_) do { static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_decode_layers<MODE_, NP_>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_decode_layers<MODE_, NP_>), dim3((unsigned) grid), dim3(1024), lds, st, a); } while (0)
    if (rope_mode == 0) { if (npre4) GOP(0, 4); else GOP(0, 1); }
    else                { if (npre4) GOP(2, 4); else GOP(2, 1); }
#undef GOP
    LAUNCH_CHECK();
    return CLLM_OK;
}
// after the stream has been synchronized: 0, or the phase whose barrier timed out
int decode_layers_error(const void * state, unsigned * phase) {
    pg_bar hb;
    HIP_TRY(hipMemcpy(&hb, state, sizeof(pg_bar), hipMemcpyDeviceToHost));
    *phase = hb.err[0];
    return CLLM_OK;
}
