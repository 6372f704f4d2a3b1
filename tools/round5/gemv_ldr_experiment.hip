// gemv_ring.hip -- k_gemv_ldr: the decode step's Q4_K mat-vec with a LOADER WAVE that keeps the weight stream running through the activation prologue.
//
// Same contract and the same arithmetic as k_gemv_dec (gemv_decode_kernel.h): the activation row is produced in the kernel (prologue 1..4), a wave owns a row,
// four lanes own a super-block per step (q4k_emit4), the reference's 8 + 4 fp32 chains run over the records in block order (q4k_chain) -- bit-identical to
// libggml-cpu.so's ggml_vec_dot_q4_K_q8_K (ggml-cpu/arch/x86/quants.c:1742-1822) behind quantize_row_q8_K (ggml-quants.c:2555-2592), which is what
// ggml_compute_forward_mul_mat does per row (ggml-cpu/ggml-cpu.c:1229-1421).  What differs is who asks for the weights:
//   * k_gemv_dec: every wave loads its own rows into VGPRs, ONE step (16 super-blocks = 2304 B) ahead.  During a launch's 2.5-3.5 us of prologue (RMS_NORM,
//     quantize_row_q8_K of the activation, redone by every workgroup) 36 KB per CU are in flight and land after ~1.5 us -- then HBM idles until the barrier opens.
//   * a deeper request burst by the same waves (round 5's first form: a per-wave LDS-DMA ring, 3 steps at kernel entry) is SLOWER: a CU holds ~50-64 KB of
//     requests in flight, a wave that issues beyond that blocks in its issue -- inside its share of the prologue (profiles/r05_ring_per_wave_prefetch.txt).
//   * here wave 15 of the workgroup is a LOADER (MI355X_MICROARCH.md rows ldsdma-fill / prefetch-credit): it has no share in the prologue, starts at kernel entry
//     and fills per-consumer rings of NS step-sized slots in LDS by LDS-DMA (global_load_lds_dwordx4, no VGPRs) as fast as the memory system takes the requests:
//     when the 15 consumer waves leave the prologue, up to 104 KB per CU (27 MB over the chip) are ON the chip or on their way, and the stream never stops
//     until the last row.  The issue order is deterministic -- round r: step r of consumers 0..14 -- so the loader's state is a handful of SGPRs (base += stride),
//     a step's global sequence number is a closed form, and ONE counter publishes what has landed (the loader retires its DMA in order: s_waitcnt vmcnt(3 x steps
//     still allowed in flight)).  A consumer waits for `landed > g(step)`, turns the slot into chain records, and publishes `consumed[c] = step + 1`; the loader
//     re-uses slot r % NS of a round once every consumer of the round has consumed step r - NS.  The consumers synchronise among themselves through an LDS counter
//     (the loader never joins a barrier after the entry one: an s_barrier would park it).  Every wait is bounded and reports through the library's error word.
// Dealing: k_gemv_dec's, over 15 consumer waves per workgroup.  Epilogues and output: k_gemv_dec's.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

#define LDR_NC    15            // consumer waves (wave 15: the loader)
#define LDR_SLOT  2304          // one step of one consumer: 16 super-blocks of 144 bytes
#define LDR_FLAGS 256           // bytes of LDS flags: word 0 = steps landed, word 1 = the consumers' barrier counter, words 16..30 = consumed[c]
#ifndef LDR_NS
#define LDR_NS    3             // ring slots per consumer
#endif
#ifndef LDR_D
#define LDR_D     19            // steps the loader keeps in flight at most (three DMA instructions each; vmcnt counts to 63)
#endif
#define LDR_SPIN  (1u << 21)    // bound of every wait (x s_sleep 1): tens of milliseconds, then the error word is set and the launch winds down

#define TS(k) do { if (ts && threadIdx.x == 0) ts[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

template <int N> __device__ __forceinline__ void ldr_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// one step: three DMA instructions, 144 x 16 bytes, global (base + lane offsets) -> LDS (dst + 16 * lane); M0 is set and restored inside the statement
__device__ __forceinline__ void ldr_dma_step(const char * base /* uniform */, unsigned a0, unsigned a1, unsigned a2, unsigned dst /* uniform */, bool third /* lane < 16 */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                 "s_add_u32 m0, %4, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(a0), "v"(a1), "s"(base), "s"(dst) : "memory", "scc");
    if (third) asm volatile("s_mov_b32 %0, m0\n\ts_add_u32 m0, %3, 0x800\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                            : "=&s"(keep) : "v"(a2), "s"(base), "s"(dst) : "memory", "scc");
}
__device__ __forceinline__ unsigned lds_load_u32(const unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store_u32(unsigned * p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// the consumers' barrier: a monotonic LDS counter (lane 0 of every consumer wave arrives, everybody polls for `target` arrivals)
__device__ __forceinline__ void ldr_consumer_barrier(unsigned * bar, unsigned target, int lane, unsigned * err) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                      // this wave's LDS stores are done (the LDS executes a wave's operations in order)
    if (lane == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned n = 0;
#pragma clang loop unroll(disable)
    while (lds_load_u32(bar) < target) { __builtin_amdgcn_s_sleep(1); if (++n > LDR_SPIN) { if (lane == 0) *err = 201; break; } }
    asm volatile("" ::: "memory");
}

template <int PRO, int EPI, int NPRE>
__global__ void __launch_bounds__(1024) k_gemv_ldr(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ W, int nblk, int kfull, int nrem, float eps,
                                                   float * __restrict__ dst, const float * __restrict__ bias, const float * resid, unsigned * err, unsigned long long * ts) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RU = EPI == 1 ? 2 : 1, NS = LDR_NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = nblk * 256;
    const int S = (nblk + 15) >> 4, RS = RU * S;
    const int b = blockIdx.x, grid = gridDim.x, nwaves = grid * LDR_NC;
    const int arb = (int) act_row_bytes(K, 256);
    unsigned * fl = (unsigned *)(lds + arb + LDR_NC * Q4K_CHAIN_BYTES);
    const int ring_off = arb + LDR_NC * Q4K_CHAIN_BYTES + LDR_FLAGS;
    const unsigned nb01 = (unsigned) nblk * 144u;
    // the schedule both sides know: consumers c < cx of this workgroup own one unit more (the last, partial round of units is dealt workgroup-interleaved: unit
    // kfull * nwaves + c * grid + b); round r < T0: step r of all 15 consumers, T0 <= r < T0 + RS: of consumers 0..cx-1
    const int T0 = kfull * RS;
    int cx = 0;
    for (int c = 0; c < LDR_NC; c++) cx += (c * grid + b < nrem) ? 1 : 0;

    if (wave == LDR_NC) {
        // ================= the loader =================
        __builtin_amdgcn_s_setprio(3);
        fl[lane] = 0u;                                                      // 64 words of flags
        lds_barrier();                                                      // the entry barrier: the only one this wave joins
        const unsigned ring0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)(lds + ring_off);
        const unsigned vo0 = (unsigned) lane * 16u, vo1 = vo0 + 1024u, vo2 = vo0 + 2048u;
        const bool third = lane < 16;
        const int R = T0 + (cx ? RS : 0);
        int ik = 0, isub = 0, is = 0, slot = 0, inflight = 0;
        unsigned landed = 0;
        // the loader retires its DMA in order with CONSTANT waits only (a variable count would be a called switch: measured, the loader then sets the pace at ~150 ns
        // per step): in the steady state the oldest of LDR_D + 1 steps in flight; otherwise -- the rings are full, or the stream has ended -- everything in flight at once
        auto retire_oldest = [&]() {
            ldr_wait_vm<3 * LDR_D>();
            inflight--; landed++;
            if (lane == 0) lds_store_u32(fl, landed);
        };
        auto flush = [&]() {
            ldr_wait_vm<0>();
            landed += (unsigned) inflight; inflight = 0;
            if (lane == 0) lds_store_u32(fl, landed);
        };
#pragma clang loop unroll(disable)
        for (int r = 0; r < R; r++) {
            const int nact = r < T0 ? LDR_NC : cx;
            if (r >= NS) {                                                  // slot r % NS of every consumer of this round must have been consumed (step r - NS)
                const unsigned need = (unsigned)(r - NS + 1);
                unsigned n = 0;
#pragma clang loop unroll(disable)
                for (;;) {
                    const unsigned cons = lane < nact ? lds_load_u32(fl + 16 + lane) : 0xffffffffu;
                    if (__ballot(cons < need) == 0ull) break;
                    if (inflight) flush(); else __builtin_amdgcn_s_sleep(1);
                    if (++n > LDR_SPIN) { if (lane == 0) *err = 202; break; }
                }
            }
            const bool full = ik < kfull;
            const unsigned unit0 = full ? (unsigned)(ik * nwaves + b * LDR_NC) : (unsigned)(kfull * nwaves + b);
            const unsigned stride = (full ? 1u : (unsigned) grid) * (unsigned) RU * nb01;
            const char * base = W + (size_t)(unit0 * (unsigned) RU + (unsigned) isub) * nb01 + (size_t)(unsigned) is * LDR_SLOT;
            unsigned dstb = ring0 + (unsigned) slot * LDR_SLOT;
            unsigned a0 = vo0, a1 = vo1, a2 = vo2;
            const int nb = nblk - 16 * is;
            if (nb < 16) {                                                  // the row's last, partial step: lanes past its end re-read its first chunk (bytes never used)
                const unsigned lim = 144u * (unsigned) nb;
                a0 = a0 < lim ? a0 : 0u; a1 = a1 < lim ? a1 : 0u; a2 = a2 < lim ? a2 : 0u;
            }
#pragma clang loop unroll(disable)
            for (int c = 0; c < nact; c++) {
                const char * ub = (const char *)(((unsigned long long)(unsigned) __builtin_amdgcn_readfirstlane((int)((unsigned long long) base >> 32)) << 32) |
                                                 (unsigned) __builtin_amdgcn_readfirstlane((int)(unsigned long long) base));
                ldr_dma_step(ub, a0, a1, a2, (unsigned) __builtin_amdgcn_readfirstlane((int) dstb), third);
                base += stride; dstb += NS * LDR_SLOT;
                if (++inflight > LDR_D) retire_oldest();
            }
            if (++slot == NS) slot = 0;
            if (++is == S) { is = 0; if (++isub == RU) { isub = 0; ik++; } }
        }
        if (inflight) flush();
        return;
    }

    // ================= the 15 consumer waves =================
    // ---- (1) this thread's activation groups: 960 threads x 4 values per pass (a wave = one 256-block), requested before anything else ----
    const float * gp = (PRO == 1 || PRO == 4) ? pw : PRO == 3 ? px + 4 : px;
    constexpr int vmul = PRO == 3 ? 2 : 1;
    constexpr int PASS = LDR_NC * 256;
    const int e0 = tid * 4;
    f32x4 vv[NPRE], gg[PRO != 2 ? NPRE : 1];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * PASS, ec = e < K ? e : 0;
        vv[u] = *(const f32x4 *)(px + ec * vmul);
        if (PRO != 2) gg[u] = *(const f32x4 *)(gp + ec * vmul);
    }
    TS(0);
    lds_barrier();                                                          // entry barrier: the flags are zero behind it
    TS(1);
    unsigned bar_target = 0;
    unsigned * bar = fl + 1;

    // ---- (2) the activation row: [RMS_NORM * weight | SiLU * up |] quantize_row_q8_K -> LDS (act layout of common.h), k_gemv_dec's arithmetic ----
    float scale = 1.0f;
    if (PRO == 1) {
        __shared__ double part[16];
        // the sum of squares as a tree (per thread in increasing index, DPP wave reduction, the wave partials in wave order); rms_scale's interval test proves per row
        // that the order cannot matter, else wave 0 redoes the sum in the reference's serial order (common.h)
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            if (e0 + u * PASS < K) { const f32x4 v = vv[u]; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        }
        sum = wave_sum_d(sum);
        if (lane == 0) part[wave] = sum;
        bar_target += LDR_NC; ldr_consumer_barrier(bar, bar_target, lane, err);
        double tot = part[0];
#pragma unroll
        for (int w = 1; w < LDR_NC; w++) tot += part[w];
        float m = rms_mean(tot, K);
        const double d = tot * ((double)(2 * (int64_t) K + 16) * 0x1p-53);
        if (!(rms_mean(tot - d, K) == rms_mean(tot + d, K))) {             // uniform over the consumers (every thread holds the same `tot`)
            bar_target += LDR_NC; ldr_consumer_barrier(bar, bar_target, lane, err);
            if (wave == 0) { const double ss = rms_serial_sumsq<false>(px, nullptr, K); if (lane == 0) part[0] = ss; }
            bar_target += LDR_NC; ldr_consumer_barrier(bar, bar_target, lane, err);
            m = rms_mean(part[0], K);
        }
        scale = 1.0f / sqrtf(m + eps);
    }
    const int nv = K & ~7;
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * PASS;
        if (e < K) {
            f32x4 v = vv[u];
            if (PRO == 3) {
                const f32x4 p0 = vv[u], p1 = gg[PRO != 2 ? u : 0];
                v.x = silu_any(p0.x, e + 0 < nv) * p0.y; v.y = silu_any(p0.z, e + 1 < nv) * p0.w;
                v.z = silu_any(p1.x, e + 2 < nv) * p1.y; v.w = silu_any(p1.z, e + 3 < nv) * p1.w;
            }
            if (PRO == 4) {
                const f32x4 g = gg[PRO != 2 ? u : 0];
                v.x = silu_any(v.x, e + 0 < nv) * g.x; v.y = silu_any(v.y, e + 1 < nv) * g.y; v.z = silu_any(v.z, e + 2 < nv) * g.z; v.w = silu_any(v.w, e + 3 < nv) * g.w;
            }
            if (PRO == 1) { const f32x4 g = gg[PRO != 2 ? u : 0]; v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w; }
            quant4_store<256, false>(lds, K, e, lane, v);
        }
    }
    TS(2);
    bar_target += LDR_NC; ldr_consumer_barrier(bar, bar_target, lane, err);
    TS(3);

    // ---- (3) the rows: wait for the step's slot, turn it into chain records, hand the slot back, advance the chains ----
    const int c = wave;
    const int total = T0 + (c < cx ? RS : 0);
    const int lin = b * LDR_NC + c, alt = c * grid + b;
    auto unit_of = [&](int k) { return k * nwaves + (k < kfull ? lin : alt); };
    const int grp = lane >> 2, j = lane & 3;
    const q4k_sel4 L = q4k_lane_sel4(lane);
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 256);
    char * chain = lds + arb + c * Q4K_CHAIN_BYTES;
    const char * myring = lds + ring_off + c * (NS * LDR_SLOT) + grp * 144;
    const int l16 = lane & 15;
    float acc = 0.0f, gate = 0.0f;
    unsigned seen = 0;
    int ck = 0, csub = 0, cs = 0, cslot = 0;
#pragma clang loop unroll(disable)
    for (int r = 0; r < total; r++) {
        const unsigned g = r < T0 ? (unsigned)(r * LDR_NC + c) : (unsigned)(T0 * LDR_NC + (r - T0) * cx + c);
        if (seen <= g) {
            unsigned n = 0;
#pragma clang loop unroll(disable)
            while ((seen = lds_load_u32(fl)) <= g) { __builtin_amdgcn_s_sleep(1); if (++n > LDR_SPIN) { if (lane == 0) *err = 203; break; } }
            asm volatile("" ::: "memory");
        }
        const char * sp = myring + cslot * LDR_SLOT;
        const u32x4 h = *(const u32x4 *) sp, qa = *(const u32x4 *)(sp + 16 + 32 * j), qb = *(const u32x4 *)(sp + 32 + 32 * j);
        const int bb = 16 * cs + grp;
        const bool ok = bb < nblk;
        q4k_emit4(h, qa, qb, lds, off_d, off_s, ok ? bb : 0, ok, L, chain);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // the slot's reads have returned (and the records are stored)
        if (lane == 0) lds_store_u32(fl + 16 + c, (unsigned)(r + 1));
        {
            wave_lds_fence();
            q4k_chain(chain, 8, l16, acc);
            wave_lds_fence();
        }
        if (++cslot == NS) cslot = 0;
        if (++cs == S) {                                                    // row complete: finish the chains, epilogue, store (lane 0)
            float v = chain_finish<1>(acc);
            const int cunit = unit_of(ck), crow = cunit * RU + csub;
            if (EPI == 1) {
                if (csub == 0) gate = v;
                else if (lane == 0) dst[cunit] = silu_poly(gate) * v;
            } else {
                if (bias)  v = v + uniform_load_f32(bias + crow);
                if (resid) v = v + uniform_load_f32(resid + crow);
                if (lane == 0) dst[crow] = v;
            }
            acc = 0.0f; cs = 0;
            if (++csub == RU) { csub = 0; ck++; }
        }
    }
    TS(4);
    if (ts) { bar_target += LDR_NC; ldr_consumer_barrier(bar, bar_target, lane, err); TS(5); }
}
#undef TS

static unsigned long long * g_ring_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_ring_ts(unsigned long long * dev_buf) { g_ring_ts = dev_buf; }   // tools only

// which launches take the loader form.  CLLM_GEMV_LDR: 0 off; 1 (default) where it measured faster than k_gemv_dec; 2 everything it can take
int gemv_ldr_mode() { static const int v = getenv("CLLM_GEMV_LDR") ? atoi(getenv("CLLM_GEMV_LDR")) : 1; return v; }

// Q4_K, K % 256 == 0, 16-byte aligned rows and vectors; CLLM_E_UNSUPPORTED: k_gemv_dec takes the launch
int launch_gemv_ldr(hipStream_t st, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                    const float * bias, const float * resid) {
    const int mode = gemv_ldr_mode();
    if (!mode || K % 256 || pro < 1 || pro > 4 || nrows <= 0 || (uint64_t) nrows * (uint64_t)(K / 256 * 144) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (K > 8 * LDR_NC * 256) return CLLM_E_UNSUPPORTED;
    if (epi != 0 && (epi != 1 || pro != 1 || nrows % 2 || (nrows / 2) % 8 || bias || resid)) return CLLM_E_UNSUPPORTED;
    if ((((uintptr_t) W) & 15) || (((uintptr_t) px) & 15) || ((pro == 1 || pro == 4) && (((uintptr_t) pw) & 15))) return CLLM_E_UNSUPPORTED;
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    const int cus = device_cu_count();
    // (mode 1) a wave needs several steps for a stream to run ahead in: one or two steps per wave are on their way at entry in either form
    if (mode == 1 && units * ((K / 256 + 15) / 16) * (epi == 1 ? 2 : 1) < 3 * 16 * (int64_t) cus) return CLLM_E_UNSUPPORTED;
    int64_t grid = (units + LDR_NC - 1) / LDR_NC;
    if (grid > cus) grid = cus;
    const int64_t nwaves = grid * LDR_NC;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / 256);
    const size_t lds = act_row_bytes(K, 256) + LDR_NC * (size_t) Q4K_CHAIN_BYTES + LDR_FLAGS + LDR_NC * (size_t) LDR_NS * LDR_SLOT;
    if (lds > 160 * 1024 - 256) return CLLM_E_UNSUPPORTED;
    unsigned * err = nullptr;
    { const int rc = kernel_error_word(&err); if (rc) return rc; }
    const int npre = K <= LDR_NC * 256 ? 1 : K <= 2 * LDR_NC * 256 ? 2 : K <= 4 * LDR_NC * 256 ? 4 : 8;
#define GOR(PRO_, EPI_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_ldr<PRO_, EPI_, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_ldr<PRO_, EPI_, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const char *) W, nblk, kfull, nrem, eps, dst, bias, resid, err, g_ring_ts); } while (0)
#define GON(PRO_, EPI_) do { if (npre == 1) GOR(PRO_, EPI_, 1); else if (npre == 2) GOR(PRO_, EPI_, 2); else if (npre == 4) GOR(PRO_, EPI_, 4); else GOR(PRO_, EPI_, 8); } while (0)
    if (pro == 1 && epi == 1) GON(1, 1);
    else if (pro == 1)        GON(1, 0);
    else if (pro == 2)        GON(2, 0);
    else if (pro == 4)        GON(4, 0);
    else                      GON(3, 0);
#undef GON
#undef GOR
    LAUNCH_CHECK();
    return CLLM_OK;
}
