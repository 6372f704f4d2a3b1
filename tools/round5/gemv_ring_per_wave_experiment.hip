// gemv_ring.hip -- k_gemv_ring: the decode step's Q4_K mat-vec with the weight stream running THROUGH the activation prologue.
//
// Same contract and the same arithmetic as k_gemv_dec (gemv_decode_kernel.h): the activation row is produced in the kernel (prologue 1..4), a wave owns a row,
// four lanes own a super-block per step (q4k_emit4), the reference's 8 + 4 fp32 chains run over the records in block order (q4k_chain) -- bit-identical to
// libggml-cpu.so's ggml_vec_dot_q4_K_q8_K (ggml-cpu/arch/x86/quants.c:1742-1822) behind quantize_row_q8_K (ggml-quants.c:2555-2592), which is what
// ggml_compute_forward_mul_mat does per row (ggml-cpu/ggml-cpu.c:1229-1421).  What differs is how the weights reach the lanes:
//   * k_gemv_dec loads them into VGPRs, ONE step (16 super-blocks = 2304 B per wave) ahead: a launch's 2.5-3.5 us of prologue (RMS_NORM, quantize_row_q8_K of
//     the activation, redone by every workgroup) pass with 36 KB per CU in flight -- HBM idles.  More steps in registers were measured slower (the 128-register
//     budget of a 1024-thread workgroup, and the loads' issue time in front of the prologue);
//   * here every wave owns a ring of NS step-sized slots in LDS, filled by LDS-DMA (global_load_lds_dwordx4: 64 x 16 bytes per instruction, no VGPRs, counted by
//     vmcnt).  The first NS steps are requested at kernel entry, right behind the activation loads: 110 KB per CU = 28 MB over the chip are in flight while the
//     prologue computes, i.e. all of qkv (14 MB) / o (9 MB) / most of down (33 MB) and the first third of gate/up are ON the chip when the prologue barrier opens.
//     A slot is refilled as soon as its step has been turned into chain records.  A wave waits for its own DMA with a counted s_waitcnt (loads retire in order),
//     so there is no cross-wave hand-off at all.
//   * the activation loads are hand-issued (asm) and waited for with vmcnt(3 NS): a compiler-counted load would be waited for with vmcnt(0), draining the ring.
// Dealing of rows to waves, epilogues and the output are k_gemv_dec's.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

#define RING_SLOT 2304          // one step of one wave: 16 super-blocks of 144 bytes

#define TS(k) do { if (ts && threadIdx.x == 0) ts[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)

// one DMA instruction: 64 (or fewer: EXEC) lanes x 16 bytes, global (base + voff) -> LDS (lds_dst + 16 * lane)
__device__ __forceinline__ void ring_dma16(const char * base /* wave-uniform */, unsigned voff, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
#ifdef RING_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void ring_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
// a 16-byte activation load the compiler does not count (see the header): the value is usable after ring_wait_vm + ring_pin
__device__ __forceinline__ f32x4 ring_load4(const float * p) { f32x4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v; }
__device__ __forceinline__ void ring_pin(f32x4 & v) { asm volatile("" : "+v"(v)); }

template <int PRO, int EPI, int NPRE, int NS>
__global__ void __launch_bounds__(1024) k_gemv_ring(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ W, int nblk, int kfull, int nrem, float eps,
                                                    float * __restrict__ dst, const float * __restrict__ bias, const float * resid, unsigned long long * ts) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int RU = EPI == 1 ? 2 : 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = nblk * 256;

    // ---- (1) this thread's activation groups, hand-issued ----
    const float * gp = (PRO == 1 || PRO == 4) ? pw : PRO == 3 ? px + 4 : px;
    constexpr int vmul = PRO == 3 ? 2 : 1;
    const int e0 = tid * 4;
    f32x4 vv[NPRE], gg[PRO != 2 ? NPRE : 1];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096, ec = e < K ? e : 0;
        vv[u] = ring_load4(px + ec * vmul);
        if (PRO != 2) gg[u] = ring_load4(gp + ec * vmul);
    }
    TS(0);

    // ---- (2) the wave's stream: its units in k_gemv_dec's dealing, RU rows each, S steps per row; the first NS steps go out now ----
    const int grp = lane >> 2, j = lane & 3;
    const int wave_in_wg = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = gridDim.x * 16;
    const int lin = blockIdx.x * 16 + wave_in_wg, alt = wave_in_wg * gridDim.x + blockIdx.x;
    const int nmine = kfull + (alt < nrem ? 1 : 0);
    const int S = (nblk + 15) >> 4;
    const int total = nmine * RU * S;
    const unsigned nb01 = (unsigned) nblk * 144u;
    auto unit_of = [&](int k) { return k * nwaves + (k < kfull ? lin : alt); };
    const int arb = (int) act_row_bytes(K, 256);
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) char *) lds + (unsigned) arb + 16u * Q4K_CHAIN_BYTES + (unsigned) wave_in_wg * (NS * RING_SLOT);
    const unsigned vo0 = (unsigned) lane * 16u, vo1 = vo0 + 1024u, vo2 = vo0 + 2048u;
    int iq = 0, ik = 0, isub = 0, is = 0, islot = 0;              // issue cursor: step ordinal, unit ordinal, row of the unit, step of the row, ring slot
    auto issue = [&]() {
        if (iq < total) {
            const unsigned row = (unsigned)(unit_of(ik) * RU + isub);
            const char * bp = W + (size_t) row * nb01 + (size_t)(unsigned) is * RING_SLOT;
            const char * ubs = (const char *)(((unsigned long long)(unsigned) __builtin_amdgcn_readfirstlane((int)((unsigned long long) bp >> 32)) << 32) |
                                              (unsigned) __builtin_amdgcn_readfirstlane((int)(unsigned long long) bp));
            const unsigned dstb = __builtin_amdgcn_readfirstlane((int)(ring + (unsigned) islot * RING_SLOT));
            const int nb = nblk - 16 * is;                         // super-blocks of this step (>= 16: a full step)
            unsigned a0 = vo0, a1 = vo1, a2 = vo2;
            if (nb < 16) {                                         // the row's last, partial step: lanes past its end re-read its first chunk (their LDS bytes are never used)
                const unsigned lim = 144u * (unsigned) nb;
                a0 = a0 < lim ? a0 : 0u; a1 = a1 < lim ? a1 : 0u; a2 = a2 < lim ? a2 : 0u;
            }
            ring_dma16(ubs, a0, dstb);
            ring_dma16(ubs, a1, dstb + 1024);
            if (lane < 16) ring_dma16(ubs, a2, dstb + 2048);
        }
        iq++;
        if (++islot == NS) islot = 0;
        if (++is == S) { is = 0; if (++isub == RU) { isub = 0; ik++; } }
    };
#pragma unroll
    for (int p = 0; p < NS; p++) issue();
    TS(1);

    // ---- (3) the activation row: [RMS_NORM * weight | SiLU * up |] quantize_row_q8_K -> LDS (act layout of common.h), k_gemv_dec's arithmetic ----
    {                                                               // the activation loads are older than the ring's requests (three per step issued): in-order return
        const int nini = total < NS ? total : NS;
        if (NS >= 3 && nini >= 3) ring_wait_vm<9>(); else if (nini == 2) ring_wait_vm<6>(); else if (nini == 1) ring_wait_vm<3>(); else ring_wait_vm<0>();
    }
#pragma unroll
    for (int u = 0; u < NPRE; u++) { ring_pin(vv[u]); if (PRO != 2) ring_pin(gg[u]); }
    float scale = 1.0f;
    if (PRO == 1) {
        __shared__ double part[16];
        // rms_block_sumsq_1024's sum (thread t: groups t, t + 1024, ... in increasing index; DPP wave reduction; the 16 wave partials in wave order) from the
        // groups already in registers, behind a barrier that does not drain the ring
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < NPRE; u++) {
            if (e0 + u * 4096 < K) { const f32x4 v = vv[u]; sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
        }
        sum = wave_sum_d(sum);
        if (lane == 0) part[tid >> 6] = sum;
        lds_barrier();
        double tot = part[0];
#pragma unroll
        for (int w = 1; w < 16; w++) tot += part[w];
        scale = rms_scale(tot, K, eps, px, nullptr, part);
    }
    const int nv = K & ~7;
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096;
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
    lds_barrier();
    TS(3);

    // ---- (4) the rows: a step's super-blocks come out of the ring, become chain records, the slot is refilled, the chains advance ----
    const q4k_sel4 L = q4k_lane_sel4(lane);
    const int off_d = (int) act_off_d(K), off_s = (int) act_off_s(K, 256);
    char * chain = lds + arb + wave_in_wg * Q4K_CHAIN_BYTES;
    const char * myring = lds + arb + 16 * Q4K_CHAIN_BYTES + wave_in_wg * (NS * RING_SLOT) + grp * 144;
    const int l16 = lane & 15;
    float acc = 0.0f, gate = 0.0f;
    int cq = 0, ck = 0, csub = 0, cs = 0, cslot = 0;                // consume cursor
    while (cq < total) {
        // slot cq has landed when at most the steps issued after it are outstanding (three DMA instructions each)
        const int later = (iq < total ? iq : total) - cq - 1;
        if (NS >= 3 && later >= 2) ring_wait_vm<6>(); else if (later >= 1) ring_wait_vm<3>(); else ring_wait_vm<0>();
        const char * sp = myring + cslot * RING_SLOT;
        const u32x4 h = *(const u32x4 *) sp, qa = *(const u32x4 *)(sp + 16 + 32 * j), qb = *(const u32x4 *)(sp + 32 + 32 * j);
        const int b = 16 * cs + grp;
        const bool ok = b < nblk;
        q4k_emit4(h, qa, qb, lds, off_d, off_s, ok ? b : 0, ok, L, chain);
        issue();                                                    // refill the slot (its reads have returned: the records computed from them are stored)
        {
            wave_lds_fence();
            q4k_chain(chain, 8, l16, acc);
            wave_lds_fence();
        }
        cq++;
        if (++cslot == NS) cslot = 0;
        if (++cs == S) {                                            // row complete: finish the chains, epilogue, store (lane 0)
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
    if (ts) { __syncthreads(); TS(5); }
}
#undef TS

static unsigned long long * g_ring_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_ring_ts(unsigned long long * dev_buf) { g_ring_ts = dev_buf; }   // tools only

// 0: off (k_gemv_dec takes the launch); 2 / 3: ring depth.  CLLM_GEMV_RING
int gemv_ring_mode() { static const int v = getenv("CLLM_GEMV_RING") ? atoi(getenv("CLLM_GEMV_RING")) : 3; return v; }

// Q4_K, K % 256 == 0, rows 16-byte aligned (144-byte blocks: always); CLLM_E_UNSUPPORTED: k_gemv_dec takes the launch
int launch_gemv_ring(hipStream_t st, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                     const float * bias, const float * resid) {
    int ns = gemv_ring_mode();
    if (ns < 2 || K % 256 || pro < 1 || pro > 4 || nrows <= 0 || (uint64_t) nrows * (uint64_t)(K / 256 * 144) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (ns > 3) ns = 3;
    if (K > ((pro == 2 || pro == 4) ? 32768 : 16384)) return CLLM_E_UNSUPPORTED;
    if (epi != 0 && (epi != 1 || pro != 1 || nrows % 2 || (nrows / 2) % 8 || bias || resid)) return CLLM_E_UNSUPPORTED;
    if ((((uintptr_t) W) & 15) || (((uintptr_t) px) & 15) || ((pro == 1 || pro == 4) && (((uintptr_t) pw) & 15))) return CLLM_E_UNSUPPORTED;
    const int64_t units = epi == 1 ? nrows / 2 : nrows;
    int64_t grid = (units + 15) / 16;
    if (grid > device_cu_count()) grid = device_cu_count();
    const int64_t nwaves = grid * 16;
    const int kfull = (int)(units / nwaves), nrem = (int)(units % nwaves), nblk = (int)(K / 256);
    const size_t fixed = act_row_bytes(K, 256) + 16 * (size_t) Q4K_CHAIN_BYTES;
    if (fixed + 16 * (size_t) ns * RING_SLOT > 160 * 1024 - 256) ns = 2;
    if (fixed + 16 * (size_t) ns * RING_SLOT > 160 * 1024 - 256) return CLLM_E_UNSUPPORTED;
    const size_t lds = fixed + 16 * (size_t) ns * RING_SLOT;
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOR(PRO_, EPI_, NPRE_, NS_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_ring<PRO_, EPI_, NPRE_, NS_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_ring<PRO_, EPI_, NPRE_, NS_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const char *) W, nblk, kfull, nrem, eps, dst, bias, resid, g_ring_ts); } while (0)
#define GON(NS_) do { \
        if (pro == 1 && epi == 1) { if (npre == 1) GOR(1, 1, 1, NS_); else GOR(1, 1, 4, NS_); } \
        else if (pro == 1)        { if (npre == 1) GOR(1, 0, 1, NS_); else GOR(1, 0, 4, NS_); } \
        else if (pro == 2)        { if (npre == 1) GOR(2, 0, 1, NS_); else if (npre == 4) GOR(2, 0, 4, NS_); else GOR(2, 0, 8, NS_); } \
        else if (pro == 4)        { if (npre == 1) GOR(4, 0, 1, NS_); else if (npre == 4) GOR(4, 0, 4, NS_); else GOR(4, 0, 8, NS_); } \
        else                      { if (npre == 1) GOR(3, 0, 1, NS_); else GOR(3, 0, 4, NS_); } } while (0)
    if (ns == 3) GON(3); else GON(2);
#undef GON
#undef GOR
    LAUNCH_CHECK();
    return CLLM_OK;
}
