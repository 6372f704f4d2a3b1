// ffn_fused.hip -- k_ffn_dec: the decode step's FFN block (BaseMLP::forward, src/layers.cpp:2475-2497, behind LMBlock1Forward's post_attention_layernorm :2744-2758) as ONE launch:
//     g = SiLU(Wgate . a) * (Wup . a),  a = quantize_row_q8_K(RMS_NORM(x) * w)          [gate/up rows interleaved: row 2u = gate_u, 2u + 1 = up_u]
//     x = Wdown . quantize_row_q8_K(g) + x
// The arithmetic is k_gemv_dec's (gemv_decode_kernel.h): a wave owns a row, four lanes a 144-byte Q4_K super-block per step (q4k_emit4), the reference's 8 + 4 fp32 chains run
// over the records in block order (q4k_chain) -- ggml_vec_dot_q4_K_q8_K's AVX2 order (ggml-cpu/arch/x86/quants.c:1742-1822) behind quantize_row_q8_K (ggml-quants.c:2555-2592),
// which is what ggml_compute_forward_mul_mat does per row (ggml-cpu/ggml-cpu.c:1229-1421): bit-identical to libggml-cpu.so, and to the two launches this one replaces.
//
// What changes is what happens BETWEEN the two mat-vecs.  As two launches the down projection pays a launch boundary (~1.7 us) and a prologue redone by every workgroup
// (quantize 14336 values: ~2.5 us) during which HBM idles -- 4.2 of its 9.8 us (profiles/r05_bench_line.json layer_split).  Here:
//   * the weight stream never stops at the edge.  Every wave owns a ring of NS step-sized slots (2304 B = 16 super-blocks) in LDS, filled by LDS-DMA (global_load_lds_dwordx4: no
//     VGPRs, counted by vmcnt, retired in order); the wave's ITEMS -- its gate/up rows, then its down row -- are one FIFO through that ring, so while a wave waits at the
//     edge its first NS steps of the down projection (6.9 of a row's 8.1 KB; 28 of the 33 MB over the chip) are already on the chip or on their way.  A launch-scoped ring lost
//     to the register path in round 5 (profiles/r05_ring_per_wave_prefetch.txt: the requests of three steps per wave at kernel ENTRY block the prologue's own loads in the
//     CU's issue queue); here one step goes out at entry, the others when the activation has landed, and the ring's credit is collected at the EDGE, not at the entry.
//   * the hand-off of g is PROGRESSIVE.  Features leave their producing wave as 8-byte {value, epoch} granules (one write-through store: data-tagged, no flag, no fence;
//     MI355X_MICROARCH.md rows handoff-1to1 / allgather).  Units are dealt in rounds of 4096 = 16 quantization blocks, so the blocks of round r are complete while round r + 1
//     streams: wave w of EVERY workgroup gathers block 16 r + w as one more item of its FIFO (three DMA instructions with the L1-bypassing policy into a ring slot, consumed
//     two items later: no extra wait), checks the tags, quantizes it (quant4_q8_K: a wave = a 256-block) into the down projection's activation row in LDS.  Only the LAST
//     round's blocks are left for the edge proper: one polled gather per wave.  The 2.5 us prologue and the boundary become one gather pass + one LDS barrier.
// Scope: Q4_K gate/up and down, H <= 4096 (one prologue group per thread), F % 256 == 0, 256 CUs (a round = 16 blocks), every wave at least one unit; anything else: the two launches.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

#define FFN_SLOT 2304           // one step of one wave: 16 super-blocks of 144 bytes (a gather item uses 2048 of it: 256 granules)
#ifndef FFN_NS
#define FFN_NS 3                // ring slots per wave
#endif
#define FFN_POLLS (1 << 14)     // bound of the edge's polling (x one memory round trip): tens of milliseconds, then the error word is set and the launch winds down

#define TS(k) do { if (ts && threadIdx.x == 0) ts[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

// one DMA instruction: 64 (or fewer: EXEC) lanes x 16 bytes, global (base + voff) -> LDS (lds_dst + 16 * lane); M0 is set and restored inside the statement
__device__ __forceinline__ void ffn_dma16(const char * base /* wave-uniform */, unsigned voff, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
// the same past the L1 (sc1: served by the L2 / fabric -- the granules were written by other CUs during this launch)
__device__ __forceinline__ void ffn_dma16_coh(const char * base, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc0 sc1\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void ffn_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// a 16-byte activation load the compiler does not count (a compiler-counted load would be waited for with vmcnt(0), draining the ring): usable after ffn_wait_vm + ffn_pin
__device__ __forceinline__ f32x4 ffn_load4(const float * p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void ffn_pin(f32x4 & v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ const char * ffn_uniform_ptr(const char * p) {
    return (const char *)(((unsigned long long)(unsigned) __builtin_amdgcn_readfirstlane((int)((unsigned long long) p >> 32)) << 32) | (unsigned) __builtin_amdgcn_readfirstlane((int)(unsigned long long) p));
}

// NULL_EDGE (tools only): the gathers take whatever the granule buffer holds -- no tag test, no polling: the launch with a hand-off that costs nothing (the bound of the design)
template <int NS, bool NULL_EDGE>
__global__ void __launch_bounds__(1024) k_ffn_dec(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ Wgu, const char * __restrict__ Wd,
                                                  int nblk_h, int nblk_f, int gu_kfull, int gu_nrem, int d_kfull, int d_nrem, float eps,
                                                  char * __restrict__ gran, unsigned * epoch_p, float * xout, const float * resid, unsigned * err, unsigned long long * ts, int flags) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = nblk_h * 256, F = nblk_f * 256;
    const int arbA = (int) act_row_bytes(H, 256), arbB = (int) act_row_bytes(F, 256);
    char * actA = lds, * actB = lds + arbA;
    char * chain = lds + arbA + arbB + wave * Q4K_CHAIN_BYTES;
    const int ring_off = arbA + arbB + 16 * Q4K_CHAIN_BYTES + wave * (NS * FFN_SLOT);
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) char *) lds + (unsigned) ring_off;
    unsigned epoch;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(epoch) : "s"(epoch_p) : "memory");

    // ---- (1) this thread's activation group and norm weights, hand-issued ----
    const int e0 = tid * 4, e0c = e0 < H ? e0 : 0;
    f32x4 vv = ffn_load4(px + e0c), gg = ffn_load4(pw + e0c);
    TS(0);

    // ---- (2) the wave's FIFO of items.  gate/up units in k_gemv_dec's dealing (full rounds: unit k nwaves + 16 b + w; the partial round workgroup-interleaved: + w grid + b),
    //          two rows each, ONE step per row (H <= 4096); behind unit k >= 1 the gather of block 16 (k - 1) + w; then the wave's last block (polled); then its down rows ----
    const int grid = gridDim.x, nwaves = grid * 16;
    const int lin = blockIdx.x * 16 + wave, alt = wave * grid + blockIdx.x;
    const int nmine = gu_kfull + (alt < gu_nrem ? 1 : 0);             // gate/up units = gather jobs of this wave
    const int dmine = d_kfull + (alt < d_nrem ? 1 : 0);               // down rows of this wave
    const int S_f = (nblk_f + 15) >> 4;                               // steps per down row
    const unsigned nb01_h = (unsigned) nblk_h * 144u, nb01_f = (unsigned) nblk_f * 144u;
    auto unit_gu = [&](int k) { return k * nwaves + (k < gu_kfull ? lin : alt); };
    auto unit_d  = [&](int k) { return k * nwaves + (k < d_kfull ? lin : alt); };
    const int n_items = 2 * nmine + (nmine > 1 ? nmine - 1 : 0) + (nmine > 0 ? 1 : 0) + dmine * S_f;
    const unsigned vo0 = (unsigned) lane * 16u, vo1 = vo0 + 1024u, vo2 = vo0 + 2048u;
    int iq = 0, islot = 0;                                            // issue cursor: item ordinal, ring slot
    int i_ph = nmine > 0 ? 0 : 2, i_k = 0, i_pos = 0, i_r = 0, i_s = 0;      // 0: gate/up units (+ gathers), 1: the last block's slot, 2: down rows
    auto dma_weights = [&](const char * bp, int nb, unsigned dstb) {  // one step: nb (<= 16: the row's last, partial step) super-blocks from bp
        unsigned a0 = vo0, a1 = vo1, a2 = vo2;
        if (nb < 16) {                                                // lanes past the row's end re-read its first chunk (their LDS bytes are never used)
            const unsigned lim = 144u * (unsigned) nb;
            a0 = a0 < lim ? a0 : 0u; a1 = a1 < lim ? a1 : 0u; a2 = a2 < lim ? a2 : 0u;
        }
        const char * ub = ffn_uniform_ptr(bp);
        ffn_dma16(ub, a0, dstb);
        ffn_dma16(ub, a1, dstb + 1024);
        if (lane < 16) ffn_dma16(ub, a2, dstb + 2048);
    };
    const bool plain_gather = (flags & 2) != 0;
    auto dma_gather = [&](int job, unsigned dstb, bool coherent) {    // 256 granules of block 16 job + w; a third instruction (16 lanes) keeps every item at three
        const char * ub = ffn_uniform_ptr(gran + (size_t)(unsigned)(16 * job + wave) * 2048u);
        if (plain_gather && !coherent) {                                           // through the XCD's L2 (invalid at launch: a line is fetched once per XCD; a line fetched too early is caught by its tags)
            ffn_dma16(ub, vo0, dstb);
            ffn_dma16(ub, vo1, dstb + 1024);
            if (lane < 16) ffn_dma16(ub, vo0, dstb + 2048);
        } else {
            ffn_dma16_coh(ub, vo0, dstb);
            ffn_dma16_coh(ub, vo1, dstb + 1024);
            if (lane < 16) ffn_dma16_coh(ub, vo0, dstb + 2048);
        }
    };
    auto issue = [&]() {
        if (iq < n_items) {
            const unsigned dstb = (unsigned) __builtin_amdgcn_readfirstlane((int)(ring + (unsigned) islot * FFN_SLOT));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the slot's previous occupant has been read
            if (i_ph == 0) {
                if (i_pos < 2) dma_weights(Wgu + (size_t)(unsigned)(unit_gu(i_k) * 2 + i_pos) * nb01_h, nblk_h, dstb);
                else dma_gather(i_k - 1, dstb, false);
                if (++i_pos == (i_k >= 1 ? 3 : 2)) { i_pos = 0; if (++i_k == nmine) i_ph = 1; }
            } else if (i_ph == 1) {
                dma_gather(nmine - 1, dstb, true);                    // (requested early only to keep every item of the FIFO at three instructions; the edge asks again)
                i_ph = 2;
            } else {
                dma_weights(Wd + (size_t)(unsigned) unit_d(i_r) * nb01_f + (size_t)(unsigned) i_s * FFN_SLOT, nblk_f - 16 * i_s, dstb);
                if (++i_s == S_f) { i_s = 0; if (++i_r == dmine) i_ph = 3; }
            }
        }
        iq++;
        if (++islot == NS) islot = 0;
    };
    issue();                                                          // one step goes out with the activation loads ...
    ffn_wait_vm<3>();                                                 // ... which are older: in-order return
    ffn_pin(vv); ffn_pin(gg);
    const bool late_fill = (flags & 1) != 0;                          // the rest of the ring behind the prologue barrier (every wave has to get its requests accepted before it can
    if (!late_fill) {                                                 // reach the barrier: 110 KB per CU hold it back by the time HBM needs for them)
#pragma unroll
        for (int p = 1; p < NS; p++) issue();
    }
    TS(1);

    // ---- (3) the gate/up activation row: RMS_NORM * weight -> quantize_row_q8_K -> LDS (k_gemv_dec's PRO 1, one group per thread) ----
    {
        __shared__ double part[16];
        __shared__ float scale_w;
        double sum = 0.0;
        if (e0 < H) { const f32x4 v = vv; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        sum = wave_sum_d(sum);
        if (lane == 0) part[wave] = sum;
        lds_barrier();
        if (tid < 64) {
            double tot = part[0];
#pragma unroll
            for (int w = 1; w < 16; w++) tot += part[w];
            const float m = rms_mean(tot, H);
            const double dl = tot * ((double)(2 * (int64_t) H + 16) * 0x1p-53);
            const bool amb = !(rms_mean(tot - dl, H) == rms_mean(tot + dl, H));
            if (tid == 0) scale_w = amb ? __int_as_float(0x7fc00000) : 1.0f / sqrtf(m + eps);
        }
        lds_barrier();
        float scale = scale_w;
        if (scale != scale) {                                         // the rare ambiguous row (uniform): the reference's serial sum by wave 0 (rms_scale's fallback)
            __syncthreads();
            if (tid < 64) { const double ss = rms_serial_sumsq<false>(px, nullptr, H); if (tid == 0) part[0] = ss; }
            __syncthreads();
            scale = 1.0f / sqrtf(rms_mean(part[0], H) + eps);
        }
        if (e0 < H) {
            f32x4 v = vv; const f32x4 g = gg;
            v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w;
            quant4_store<256, false>(actA, H, e0, lane, v);
        }
    }
    TS(2);
    lds_barrier();
    TS(3);
    if (late_fill) {
#pragma unroll
        for (int p = 1; p < NS; p++) issue();
    }

    // ---- (4) the items ----
    const int grp = lane >> 2, j = lane & 3;
    const q4k_sel4 L = q4k_lane_sel4(lane);
    const int l16 = lane & 15;
    const char * myring = lds + ring_off;
    int cq = 0, cslot = 0;                                            // consume cursor
    auto wait_item = [&]() {                                          // item cq has landed when at most the items issued after it are outstanding (three DMA instructions each)
        const int later = (iq < n_items ? iq : n_items) - cq - 1;
        if (NS >= 4 && later >= 3) ffn_wait_vm<9>(); else if (NS >= 3 && later >= 2) ffn_wait_vm<6>(); else if (later >= 1) ffn_wait_vm<3>(); else ffn_wait_vm<0>();
    };
    auto next_slot = [&]() { cq++; if (++cslot == NS) cslot = 0; };
    // one weight step out of the ring: chain records from the slot's super-blocks, the slot refilled, the chains advanced
    auto step = [&](const char * act, int K, int bb0, int nblk, float & acc) {
        const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 256);
        wait_item();
        const char * sp = myring + cslot * FFN_SLOT + grp * 144;
        const u32x4 h = *(const u32x4 *) sp, qa = *(const u32x4 *)(sp + 16 + 32 * j), qb = *(const u32x4 *)(sp + 32 + 32 * j);
        const int b = bb0 + grp;
        const bool ok = b < nblk;
        q4k_emit4(h, qa, qb, act, off_d, off_s, ok ? b : 0, ok, L, chain);
        next_slot();
        issue();
        wave_lds_fence();
        q4k_chain(chain, 8, l16, acc);
        wave_lds_fence();
    };
    // the 256 granules of a block out of the current slot: tags tested, values quantized into the down projection's activation row.  false: not all of them carry this launch's epoch
    auto block_take = [&](int job, const u32x4 a, const u32x4 c) -> bool {
        const bool mine = NULL_EDGE || (a.y == epoch && a.w == epoch && c.y == epoch && c.w == epoch);
        if (__ballot(!mine) != 0ull) return false;
        const f32x4 v = { __uint_as_float(a.x), __uint_as_float(a.z), __uint_as_float(c.x), __uint_as_float(c.z) };
        quant4_store<256, false>(actB, F, (16 * job + wave) * 256 + 4 * lane, lane, v);
        return true;
    };
    auto gather_take = [&](int job) -> bool {
        const char * sp = myring + cslot * FFN_SLOT + 32 * lane;
        return block_take(job, *(const u32x4 *) sp, *(const u32x4 *)(sp + 16));
    };
    unsigned pend = 0u;                                               // gather jobs whose granules were not all there when their item came up (a slow CU): redone at the edge
    float gate = 0.0f;
    for (int ck = 0; ck < nmine; ck++) {
        const int unit = unit_gu(ck);
        float acc = 0.0f;
        step(actA, H, 0, nblk_h, acc);
        gate = chain_finish<1>(acc);
        acc = 0.0f;
        step(actA, H, 0, nblk_h, acc);
        const float up = chain_finish<1>(acc);
        const float gv = silu_poly(gate) * up;
        if (lane == 0) __hip_atomic_store((unsigned long long *)(gran + (size_t)(unsigned) unit * 8u), ((unsigned long long) epoch << 32) | (unsigned long long) __float_as_uint(gv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ck >= 1) {
            wait_item();
            if (!gather_take(ck - 1)) pend |= 1u << (ck - 1);
            next_slot();
            issue();
        }
    }
    TS(4);
    // ---- the edge: this wave's last block (and whatever was deferred), polled in the FIFO slot reserved for it ----
    if (nmine > 0) {
        wait_item();                                                  // (the early request of this item: its instructions are retired)
        pend |= 1u << (nmine - 1);
        const unsigned dstb = (unsigned) __builtin_amdgcn_readfirstlane((int)(ring + (unsigned) cslot * FFN_SLOT));
        int polls = 0;
        while (pend) {
            const int job = __builtin_ctz(pend);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            dma_gather(job, dstb, true);
            ffn_wait_vm<0>();
            if (gather_take(job)) pend &= pend - 1;
            else if (++polls > FFN_POLLS) { if (lane == 0) *err = 300u + (unsigned) job; break; }
            else __builtin_amdgcn_s_sleep(8);
        }
        next_slot();
        issue();
    }
    TS(5);
    lds_barrier();                                                    // the down projection's activation row is complete (every wave's blocks)
    TS(6);
    // ---- the down rows ----
    for (int ck = 0; ck < dmine; ck++) {
        const int row = unit_d(ck);
        float acc = 0.0f;
        for (int s = 0; s < S_f; s++) step(actB, F, 16 * s, nblk_f, acc);
        float v = chain_finish<1>(acc);
        if (resid) v = v + uniform_load_f32(resid + row);
        if (lane == 0) xout[row] = v;
    }
    TS(7);
    if (blockIdx.x == 0 && tid == 0) *epoch_p = epoch + 1u;           // (every workgroup read the word before workgroup 0 could pass the edge: each of them owns units)
}
#undef TS

static unsigned long long * g_ffn_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_ffn_ts(unsigned long long * dev_buf) { g_ffn_ts = dev_buf; }   // tools only
// 0 (default): off (the two launches -- the fused launch measured 44.5 us per block against 24.0: profiles/r06_ffn_fused_persistent_bound.txt); 1: on (opt-in, bit-identical);
// 2 (tools): the null hand-off -- NOT a correct computation
static int g_ffn_mode = -1;
int ffn_fused_mode() { if (g_ffn_mode < 0) g_ffn_mode = getenv("CLLM_FFN_FUSED") ? atoi(getenv("CLLM_FFN_FUSED")) : 0; return g_ffn_mode; }
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_ffn_fused(int mode) { g_ffn_mode = mode; }                     // tests: both paths inside one process

static unsigned long long g_ffn_launches = 0;
extern "C" __attribute__((visibility("default"))) int cllm_debug_ffn_fused_launches(void) { return (int) g_ffn_launches; }                // tests: the fused launch was really taken
// tools: CLLM_FFN_FLAGS bit 0: fill the ring behind the prologue barrier; bit 1: the mid-stream gathers through the L2
static int g_ffn_flags = -1;
static int ffn_flags() { if (g_ffn_flags < 0) g_ffn_flags = getenv("CLLM_FFN_FLAGS") ? atoi(getenv("CLLM_FFN_FLAGS")) : 1; return g_ffn_flags; }
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_ffn_flags(int f) { g_ffn_flags = f; }
size_t ffn_fused_state_bytes(int64_t F) { return 256 + (size_t) F * 8; }      // the epoch word (its own line) + the granules
// state: ffn_fused_state_bytes(F) of device memory, zeroed once by the caller (epoch 0 is replaced by 1 at the first launch).  Wgu: gate/up rows interleaved.
// x (= resid) may be xout.  CLLM_E_UNSUPPORTED (nothing launched): the caller issues the two launches.
int launch_ffn_fused(hipStream_t st, const void * Wgu, const void * Wd, int64_t H, int64_t F, const float * x, const float * norm_w, float eps, void * state, bool * state_ready, float * xout) {
    const int mode = ffn_fused_mode();
    const int cus = device_cu_count();
    if (cus != 256 || H % 256 || F % 256 || H > 4096 || F < 4096 || F > 16384 || !state) return CLLM_E_UNSUPPORTED;      // (at most four rounds of units: four gather jobs per wave)
    if ((((uintptr_t) Wgu) & 15) || (((uintptr_t) Wd) & 15) || (((uintptr_t) x) & 15) || (((uintptr_t) norm_w) & 15)) return CLLM_E_UNSUPPORTED;
    if ((uint64_t)(2 * F) * (uint64_t)(H / 256 * 144) >= (1ull << 32) || (uint64_t) H * (uint64_t)(F / 256 * 144) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    const int grid = cus, nwaves = grid * 16;
    const size_t lds = act_row_bytes(H, 256) + act_row_bytes(F, 256) + 16 * (size_t) Q4K_CHAIN_BYTES + 16 * (size_t) FFN_NS * FFN_SLOT;
    if (lds > 160 * 1024 - 512) return CLLM_E_UNSUPPORTED;
    unsigned * err = nullptr;
    { const int rc = kernel_error_word(&err); if (rc) return rc; }
    if (!*state_ready) {
        HIP_TRY(hipMemsetAsync(state, 0, ffn_fused_state_bytes(F), st));
        const unsigned one = 1u;
        HIP_TRY(hipMemcpyAsync(state, &one, 4, hipMemcpyHostToDevice, st));
        HIP_TRY(hipStreamSynchronize(st));
        *state_ready = true;
    }
    const int gu_kfull = (int)(F / nwaves), gu_nrem = (int)(F % nwaves), d_kfull = (int)(H / nwaves), d_nrem = (int)(H % nwaves);
    unsigned * epoch = (unsigned *) state;
    char * gran = (char *) state + 256;
#define GOF(NULL_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_ffn_dec<FFN_NS, NULL_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_ffn_dec<FFN_NS, NULL_>), dim3((unsigned) grid), dim3(1024), lds, st, x, norm_w, (const char *) Wgu, (const char *) Wd, (int)(H / 256), (int)(F / 256), \
                           gu_kfull, gu_nrem, d_kfull, d_nrem, eps, gran, epoch, xout, x, err, g_ffn_ts, ffn_flags()); } while (0)
    if (mode == 2) GOF(true); else GOF(false);
#undef GOF
    LAUNCH_CHECK();
    g_ffn_launches++;
    return CLLM_OK;
}

// ---- C ABI ----
// The FFN block of ONE token as one launch (include/chatllm_hip.h).  state: cllm_ffn_fused_state_bytes(F) of device memory, zero-filled by the caller before its first use.
extern "C" CLLM_API size_t cllm_ffn_fused_state_bytes(int64_t F) { return ffn_fused_state_bytes(F); }
extern "C" CLLM_API int cllm_op_ffn_fused(void * stream, const cllm_tensor * w_gate_up, const cllm_tensor * w_down, const float * x, const float * norm_w, float eps, void * state, float * xout) {
    if (!w_gate_up || !w_down || !x || !norm_w || !state || !xout) FAIL(CLLM_E_INVALID, "ffn_fused: null");
    if (w_gate_up->type != CLLM_TYPE_Q4_K || w_down->type != CLLM_TYPE_Q4_K) return CLLM_E_UNSUPPORTED;
    const int64_t H = w_gate_up->ne[0], F = w_down->ne[0];
    if (w_gate_up->ne[1] != 2 * F || w_down->ne[1] != H || !t_is_contiguous(w_gate_up) || !t_is_contiguous(w_down)) FAIL(CLLM_E_INVALID, "ffn_fused: gate/up [H, 2 F] interleaved rows, down [F, H]");
    bool ready = true;                                                   // (the caller zero-filled it: epoch 0 would match the zero tags -- the first launch must see 1)
    unsigned e = 0;
    HIP_TRY(hipMemcpyAsync(&e, state, 4, hipMemcpyDeviceToHost, (hipStream_t) stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t) stream));
    if (e == 0) ready = false;
    return launch_ffn_fused((hipStream_t) stream, w_gate_up->data, w_down->data, H, F, x, norm_w, eps, state, &ready, xout);
}
// measurement helper (tools/ffn_bench.py): the FFN block `iters` times over n_w copies of its weights (past the Infinity Cache), as the fused launch (fused = 1 / 2: null
// hand-off) or as the two launches it replaces (0); x is updated in place every time
extern "C" CLLM_API int cllm_bench_ffn(void * stream, void * const * wgu, void * const * wd, int n_w, int64_t H, int64_t F, float * x, const float * norm_w, float eps,
                                       float * g, void * state, int fused, int iters, float * avg_us) {
    if (!wgu || !wd || n_w <= 0 || iters <= 0 || !avg_us || !x || !norm_w || !g || !state) FAIL(CLLM_E_INVALID, "bench_ffn: arguments");
    hipStream_t st = (hipStream_t) stream;
    const int keep = g_ffn_mode;
    bool ready = false;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0)); HIP_TRY(hipEventCreate(&e1));
    int rc = CLLM_OK;
    for (int pass = 0; pass < 2 && !rc; pass++) {
        if (pass == 1) HIP_TRY(hipEventRecord(e0, st));
        const int n = pass == 0 ? (n_w < 4 ? n_w : 4) : iters;
        for (int i = 0; i < n && !rc; i++) {
            if (fused) { g_ffn_mode = fused; rc = launch_ffn_fused(st, wgu[i % n_w], wd[i % n_w], H, F, x, norm_w, eps, state, &ready, x); g_ffn_mode = keep; }
            else {
                rc = launch_gemv_decode(st, CLLM_TYPE_Q4_K, wgu[i % n_w], H, 2 * F, 1, x, norm_w, eps, 1, g, nullptr, nullptr);
                if (!rc) rc = launch_gemv_decode(st, CLLM_TYPE_Q4_K, wd[i % n_w], F, H, 2, g, nullptr, 0.0f, 0, x, nullptr, x);
            }
        }
    }
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e1, st));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    (void) hipEventDestroy(e0); (void) hipEventDestroy(e1);
    *avg_us = ms * 1e3f / (float) iters;
    return CLLM_OK;
}
