// gemv_rows.hip -- the decode step's Q4_K mat-vec, second form: weights staged through LDS, one lane group per ROW.
//
// Same contract as k_gemv_dec (gemv_decode.hip): activation produced in the kernel (prologue 1..4), dst = W . act (+ bias) (+ resid) or the
// SiLU(gate) * up epilogue, every row accumulated in the ORDER of the reference's AVX2 ggml_vec_dot_q4_K_q8_K (q4k.h) -- bit-identical
// to libggml-cpu.so.  What differs is who does what:
//   * k_gemv_dec gives a row to a whole wave (8 lane groups = 8 super-blocks per step): the loads coalesce by themselves, but the
//     reference's 8 + 4 serial fp32 chains then have to be fed from 64 lanes -- a reduce-scatter of the integer sums, chain records
//     through LDS -- and the VALU work per byte nearly doubles (MI355X issues a wave64 VALU instruction in 4 cycles: the kernel was at
//     the VALU limit, not at HBM's).
//   * here the weights travel HBM -> LDS by DMA (global_load_lds_dwordx4: 64 x 16 bytes per instruction, any per-lane address, no VGPRs,
//     counted by vmcnt), so the lane <-> data assignment is free: a group of 8 lanes owns one ROW and lane j IS the reference's lane
//     A(j): it reads dword A of each of the four 32-byte chunks of a super-block (ds_read2_b32), forms sumi[A] directly and keeps acc[A]
//     in a register for the whole row; lanes 0..3 of the group also carry acc_m[k].  No scatter, no records, no cross-lane traffic until
//     the row ends (5 DPP adds).  A wave streams RPW = 8 consecutive rows (RPW = 4: 16 lanes per row, two consecutive super-blocks per
//     step, the chain hops once through DPP row_ror:8), one 144-byte super-block per group and step.
//   * DMA unit ("slot") = two steps = 2304 bytes = RPW rows x 2304 / RPW contiguous bytes each (288 B = 2.25 cache lines for RPW 8), laid out
//     [row][bytes] in a ring of NS slots per wave; the consumer waits with s_waitcnt vmcnt(3 x slots issued later) -- loads retire in
//     order, so counting only our own DMA instructions is safe whatever else the compiler has in flight (it can only make the wait stricter).
// Units of RPW rows are dealt CU-interleaved (unit u -> workgroup u % grid, wave (u / grid) % 16), so every CU gets the same share.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"

#define ROWS_SLOT_BYTES 2304
#ifndef ROWS_NS
#define ROWS_NS 3
#endif

__device__ __forceinline__ float silu_poly_r(float x) { return x / (1.0f + ggml_expf_poly(0.0f - x)); }
__device__ __forceinline__ float silu_any_r(float x, bool body) { return body ? silu_poly_r(x) : x / (1.0f + libm_expf(-x)); }

// one DMA instruction: 64 (or fewer: EXEC) lanes x 16 bytes, global (base + voff) -> LDS (lds_dst + 16 * lane)
__device__ __forceinline__ void dma16(const char * base /* wave-uniform */, unsigned voff, unsigned lds_dst /* wave-uniform */) {
    unsigned keep;
#ifdef ROWS_NT
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
#else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff), "s"(base), "s"(lds_dst) : "memory");
#endif
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int PRO, int EPI, int NPRE, int RPW>
__global__ void __launch_bounds__(1024) k_gemv_rows(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ W, int nblk, int nunits, float eps,
                                                    float * __restrict__ dst, const float * __restrict__ bias, const float * resid) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int GPR = 8 / RPW;                        // lane groups (= super-blocks per step) per row
    constexpr int CPR = ROWS_SLOT_BYTES / RPW / 16;     // 16-byte chunks per row and slot (18 / 36)
    constexpr int BPS = 2 * GPR;                        // super-blocks per row and slot
    constexpr int NS = ROWS_NS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = nblk * 256;
    const unsigned nb01 = (unsigned) nblk * 144u;

    // ---- (1) this thread's activation groups: loads issued before anything else (as k_gemv_dec) ----
    const float * gp = (PRO == 1 || PRO == 4) ? pw : PRO == 3 ? px + 4 : px;
    constexpr int vmul = PRO == 3 ? 2 : 1;
    const int e0 = tid * 4;
    f32x4 vv[NPRE], gg[NPRE];
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096, ec = e < K ? e : 0;
        vv[u] = *(const f32x4 *)(px + ec * vmul);
        if (PRO != 2) gg[u] = *(const f32x4 *)(gp + ec * vmul);
    }

    // ---- (2) this wave's stream of slots: units (k * 16 + wave) * grid + block, SPU slots each; the first NS slots fly during the prologue ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int SPU = nblk / BPS;                                               // slots per unit (nblk % BPS == 0: the launcher checks)
    const int ustride = 16 * gridDim.x, u0 = wave * gridDim.x + blockIdx.x;
    const int nmine = u0 < nunits ? (nunits - u0 + ustride - 1) / ustride : 0;
    const int total = nmine * SPU;                                            // slots of this wave
    const unsigned ring = (unsigned)(size_t)(__attribute__((address_space(3))) char *) lds + (unsigned) act_row_bytes(K, 256) + (unsigned) wave * (NS * ROWS_SLOT_BYTES);
    // chunk t = 64 i + lane of a slot -> row t / CPR, 16-byte part t % CPR: byte offset inside the unit, without the slot's own advance
    unsigned goff[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { const unsigned t = 64u * i + lane; goff[i] = (t / CPR) * nb01 + (t % CPR) * 16u; }
    int iq = 0, iu = 0, is = 0;                                               // issue cursor: slot ordinal, unit ordinal, slot of the unit
    auto issue = [&]() {
        if (iq < total) {
            const char * ub = W + (size_t)(unsigned)(u0 + iu * ustride) * (size_t)(RPW * nb01) + (size_t)(unsigned) is * (CPR * 16u);
            const char * ubs = (const char *)(((unsigned long long)(unsigned) __builtin_amdgcn_readfirstlane((int)((unsigned long long) ub >> 32)) << 32) |
                                              (unsigned) __builtin_amdgcn_readfirstlane((int)(unsigned long long) ub));
            const unsigned dstb = __builtin_amdgcn_readfirstlane(ring + (unsigned)(iq % NS) * ROWS_SLOT_BYTES);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // the slot's previous occupant has been read
#ifndef ROWS_NODMA
            dma16(ubs, goff[0], dstb);
            dma16(ubs, goff[1], dstb + 1024);
            if (lane < 16) dma16(ubs, goff[2], dstb + 2048);
#endif
        }
        iq++;
        if (++is == SPU) { is = 0; iu++; }
    };
#pragma unroll
    for (int p = 0; p < NS; p++) issue();

    // ---- (3) the activation row: [RMS_NORM * weight | SiLU * up |] quantize -> LDS (act layout of common.h), exactly as k_gemv_dec ----
    float scale = 1.0f;
    if (PRO == 1) {
        __shared__ double part[16];
        const double sum = NPRE == 1 ? rms_block_sumsq_1024_one(vv[0], e0 < K, part) : rms_block_sumsq_1024(px, K, vv[0], part);
        scale = rms_scale(sum, K, eps, px, nullptr, part);
    }
    const int nv = K & ~7;
#pragma unroll
    for (int u = 0; u < NPRE; u++) {
        const int e = e0 + u * 4096;
        if (e < K) {
            f32x4 v = vv[u];
            if (PRO == 3) {
                const f32x4 p0 = vv[u], p1 = gg[u];
                v.x = silu_any_r(p0.x, e + 0 < nv) * p0.y; v.y = silu_any_r(p0.z, e + 1 < nv) * p0.w;
                v.z = silu_any_r(p1.x, e + 2 < nv) * p1.y; v.w = silu_any_r(p1.z, e + 3 < nv) * p1.w;
            }
            if (PRO == 4) {
                const f32x4 g = gg[u];
                v.x = silu_any_r(v.x, e + 0 < nv) * g.x; v.y = silu_any_r(v.y, e + 1 < nv) * g.y; v.z = silu_any_r(v.z, e + 2 < nv) * g.z; v.w = silu_any_r(v.w, e + 3 < nv) * g.w;
            }
            if (PRO == 1) { const f32x4 g = gg[u]; v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w; }
            quant4_store<256, false>(lds, K, e, lane, v);
        }
    }
    __syncthreads();
    if (EPI != 2 && nmine == 0) return;                               // (EPI 2: every wave meets the barriers of the arg-max tail)

    // ---- (4) stream the rows ----
    const int j = lane & 7;
    const int A = 4 * (j & 1) + (j & 2) + (j >> 2);                   // the reference's lane this lane is (slot order [A0 A4 A2 A6 | A1 A5 A3 A7]: chain_finish)
    const int kk = ((j & 1) << 1) | ((j >> 1) & 1);                   // mins pair of this lane (j & 3 -> [0, 2, 1, 3]); lanes 4..7 duplicate
    const int rowg = RPW == 8 ? lane >> 3 : lane >> 4;                // row of the unit
    const int gg2 = RPW == 8 ? 0 : (lane >> 3) & 1;                   // which of the row's two super-blocks per step (RPW 4)
    const bool hi = gg2 != 0;
    const unsigned lrow = ring + (unsigned) rowg * (CPR * 16u) + (unsigned) gg2 * 144u + 16u + 4u * A;       // + slot, + step: this lane's dword of chunk 0
    const unsigned lhdr = ring + (unsigned) rowg * (CPR * 16u) + (unsigned) gg2 * 144u;
    const char * act = lds;
    const float * actd = (const float *)(lds + act_off_d(K));
    const int * acts = (const int *)(lds + act_off_s(K, 256));
    typedef __attribute__((address_space(3))) const uint32_t * lptr;
    typedef __attribute__((address_space(3))) const u32x4 * lptr4;

    float acc = 0.0f, accm = 0.0f;
    float bestv = -INFINITY; int besti = 0x7fffffff;                  // EPI 2: the largest value this lane has stored and its row (first maximum wins)
    int cq = 0, cu = 0, cs = 0;                                       // consume cursor: slot ordinal, unit ordinal, slot of the unit
    while (cq < total) {
        // wait for slot cq: slots issued after it = min(iq, total) - cq - 1, three DMA instructions each
        const int later = (iq < total ? iq : total) - cq - 1;
        if (later >= 2) wait_vm<6>(); else if (later == 1) wait_vm<3>(); else wait_vm<0>();
        const unsigned sb = (unsigned)(cq % NS) * ROWS_SLOT_BYTES;
#ifndef ROWS_NOCOMPUTE
#pragma unroll
        for (int s = 0; s < 2; s++) {
            const int b = cs * BPS + s * GPR + gg2;                   // super-block of the row
            const unsigned o = sb + (unsigned) s * (GPR * 144u);
            const u32x4 h = *(lptr4)(size_t)(lhdr + o);
            lptr qp = (lptr)(size_t)(lrow + o);
            const uint32_t q0 = qp[0], q1 = qp[8], q2 = qp[16], q3 = qp[24];
            const uint32_t * ap = (const uint32_t *)(act + b * 256 + 4 * A);
            const uint32_t al0 = ap[0], ah0 = ap[8], al1 = ap[16], ah1 = ap[24], al2 = ap[32], ah2 = ap[40], al3 = ap[48], ah3 = ap[56];
            const float yd = actd[b];
            const int S0 = acts[b * 8 + 2 * kk], S1 = acts[b * 8 + 2 * kk + 1];
            const float d = h2f((uint16_t)(h.x & 0xffff)), dmin = h2f((uint16_t)(h.x >> 16));
            const uint32_t u0s = h.y & 0x3f3f3f3fu, u2s = h.z & 0x3f3f3f3fu;
            const uint32_t u1s = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
            const uint32_t u3s = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
            int t;
            t = __mul24((int)(u0s & 0xff), dot4(q0 & 0x0f0f0f0fu, al0, 0));
            t += __mul24((int)((u0s >> 8) & 0xff), dot4((q0 >> 4) & 0x0f0f0f0fu, ah0, 0));
            t += __mul24((int)((u0s >> 16) & 0xff), dot4(q1 & 0x0f0f0f0fu, al1, 0));
            t += __mul24((int)(u0s >> 24), dot4((q1 >> 4) & 0x0f0f0f0fu, ah1, 0));
            t += __mul24((int)(u1s & 0xff), dot4(q2 & 0x0f0f0f0fu, al2, 0));
            t += __mul24((int)((u1s >> 8) & 0xff), dot4((q2 >> 4) & 0x0f0f0f0fu, ah2, 0));
            t += __mul24((int)((u1s >> 16) & 0xff), dot4(q3 & 0x0f0f0f0fu, al3, 0));
            t += __mul24((int)(u1s >> 24), dot4((q3 >> 4) & 0x0f0f0f0fu, ah3, 0));
            const uint32_t mp = (kk >= 2 ? u3s : u2s) >> (16 * (kk & 1));
            const int pm = __mul24((int)(mp & 0xff), S0) + __mul24((int)((mp >> 8) & 0xff), S1);
            const float dd = yd * d, dm = (-yd) * dmin, x = (float) t, xm = (float) pm;
            if (GPR == 1) { acc = __builtin_fmaf(dd, x, acc); accm = __builtin_fmaf(dm, xm, accm); }
            else {      // two super-blocks of the row per step: the low group's chain value hops to the high group and back (DPP, 16-lane rows)
                const float a1 = __builtin_fmaf(dd, x, acc), m1 = __builtin_fmaf(dm, xm, accm);                       // valid in the low group
                const float a2 = __builtin_fmaf(dd, x, dpp_f<DPP_ROW_ROR8>(a1)), m2 = __builtin_fmaf(dm, xm, dpp_f<DPP_ROW_ROR8>(m1));     // valid in the high group
                const float a2r = dpp_f<DPP_ROW_ROR8>(a2), m2r = dpp_f<DPP_ROW_ROR8>(m2);      // (outside the select: a DPP read needs its source lanes active)
                acc = hi ? a2 : a2r; accm = hi ? m2 : m2r;                     // both groups hold the row's chain value
            }
        }
#endif
        issue();                                                      // refill the slot just consumed
        cq++;
        if (++cs == SPU) {                                            // RPW rows complete
            float hsum = acc;
            hsum = hsum + dpp_f<DPP_QUAD_XOR1>(hsum); hsum = hsum + dpp_f<DPP_QUAD_XOR2>(hsum); hsum = hsum + dpp_f<DPP_HALF_MIRROR>(hsum);
            float ms = accm;                                          // lanes j & 3 = [m0 m2 m1 m3]
            ms = ms + dpp_f<DPP_QUAD_XOR1>(ms); ms = ms + dpp_f<DPP_QUAD_XOR2>(ms);
            float v = hsum + ms;
            const int unit = u0 + cu * ustride;
            if (EPI == 1) {                                           // rows alternate gate_u, up_u
                const float up = RPW == 8 ? dpp_f<DPP_ROW_ROR8>(v) : __shfl_down(v, 16, 64);
                const bool st = RPW == 8 ? (lane & 15) == 0 : (lane & 31) == 0;
                if (st) dst[unit * (RPW / 2) + (RPW == 8 ? lane >> 4 : lane >> 5)] = silu_poly_r(v) * up;
            } else {
                const int row = unit * RPW + rowg;
                const bool st = RPW == 8 ? j == 0 : (lane & 15) == 0;
                if (EPI != 2 && (bias || resid)) {                    // the unit's RPW values through the scalar cache: their own counter, no wait on the DMA stream
                    float bsel = 0.0f, rsel = 0.0f;
#pragma unroll
                    for (int q = 0; q < RPW; q++) {
                        if (bias)  { const float x = uniform_load_f32(bias  + (size_t) unit * RPW + q); bsel = rowg == q ? x : bsel; }
                        if (resid) { const float x = uniform_load_f32(resid + (size_t) unit * RPW + q); rsel = rowg == q ? x : rsel; }
                    }
                    if (bias)  v = v + bsel;
                    if (resid) v = v + rsel;
                }
                if (st) dst[row] = v;
                if (EPI == 2 && st && (v > bestv || (v == bestv && row < besti))) { bestv = v; besti = row; }
            }
            acc = 0.0f; accm = 0.0f; cs = 0; cu++;
        }
    }
    if constexpr (EPI == 2) {
        // the greedy sampler's first stage (k_argmax_partial, decode_fused.hip) on the values still in registers: (value, row) of this workgroup's largest logit, lowest row
        // on ties, NaN never wins -- a total order, so the partition of the rows over lanes, waves and workgroups does not matter.  bias / resid carry the two output arrays.
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bestv, o, 64); const int oi = __shfl_xor(besti, o, 64);
            if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; }
        }
        __syncthreads();                                              // every wave has consumed its last slot and the activation row: the LDS is free
        float * lv = (float *) lds; int * li = (int *)(lds + 64);
        if (lane == 0) { lv[wave] = bestv; li[wave] = besti; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; w++) { const float ov = lv[w]; const int oi = li[w]; if (ov > bestv || (ov == bestv && oi < besti)) { bestv = ov; besti = oi; } }
            ((float *) bias)[blockIdx.x] = bestv; ((int *) resid)[blockIdx.x] = besti;
        }
    }
}

// K % (256 * 2 * (8 / RPW)) == 0, nrows % RPW == 0, rows 16-byte aligned; CLLM_E_UNSUPPORTED: k_gemv_dec takes the launch
static int gemv_rows_go(hipStream_t st, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                        const float * bias, const float * resid, int * grid_out) {
    // 0: off; 1 (default): only where it measured faster than k_gemv_dec -- many rows per CU (lm_head: 78 -> 62 us; gate/up and the small
    // projections are a draw or slower: one unit per wave leaves no steady state); 2: everything it can take; 8 / 4: that too, with RPW forced
    static const int mode = getenv("CLLM_GEMV_ROWS") ? atoi(getenv("CLLM_GEMV_ROWS")) : 1;
    if (!mode || K % 256 || pro < 1 || pro > 4 || nrows <= 0 || (uint64_t) nrows * (uint64_t)(K / 256 * 144) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (K > ((pro == 2 || pro == 4) ? 32768 : 16384)) return CLLM_E_UNSUPPORTED;
    if (epi == 1 && (pro != 1 || bias || resid)) return CLLM_E_UNSUPPORTED;
    if (epi == 2 && (pro != 1 || !bias || !resid)) return CLLM_E_UNSUPPORTED;
    const int nblk = (int)(K / 256), cus = device_cu_count();
    if (mode == 1 && nrows / 8 < 32 * (int64_t) cus) return CLLM_E_UNSUPPORTED;
    int rpw = mode == 8 || mode == 4 ? mode : (nrows / 8 >= 8 * (int64_t) cus ? 8 : 4);      // RPW 8 from 8 waves per CU on
    if (rpw == 8 && (nblk % 2 || nrows % 8)) rpw = 4;
    if (rpw == 4 && (nblk % 4 || nrows % 4)) { if (nblk % 2 == 0 && nrows % 8 == 0) rpw = 8; else return CLLM_E_UNSUPPORTED; }
    if (epi == 1 && nrows % (2 * rpw)) return CLLM_E_UNSUPPORTED;
    const int nunits = (int)(nrows / rpw);
    int grid = nunits < cus ? nunits : cus;
    if (epi == 2 && grid > 256) return CLLM_E_UNSUPPORTED;          // (the partial arrays hold 256 entries)
    const size_t lds = act_row_bytes(K, 256) + 16 * (size_t) ROWS_NS * ROWS_SLOT_BYTES;
    if (lds > 159 * 1024) return CLLM_E_UNSUPPORTED;      // (+ the prologue's static 128 bytes)
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOR(PRO_, EPI_, NPRE_, RPW_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_rows<PRO_, EPI_, NPRE_, RPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_rows<PRO_, EPI_, NPRE_, RPW_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const char *) W, nblk, nunits, eps, dst, bias, resid); } while (0)
#define GOP(RPW_) do { \
        if (pro == 1 && epi == 1) { if (npre == 1) GOR(1, 1, 1, RPW_); else GOR(1, 1, 4, RPW_); } \
        else if (pro == 1 && epi == 2) { if (npre == 1) GOR(1, 2, 1, RPW_); else GOR(1, 2, 4, RPW_); } \
        else if (pro == 1)        { if (npre == 1) GOR(1, 0, 1, RPW_); else GOR(1, 0, 4, RPW_); } \
        else if (pro == 2)        { if (npre == 1) GOR(2, 0, 1, RPW_); else if (npre == 4) GOR(2, 0, 4, RPW_); else GOR(2, 0, 8, RPW_); } \
        else if (pro == 4)        { if (npre == 1) GOR(4, 0, 1, RPW_); else if (npre == 4) GOR(4, 0, 4, RPW_); else GOR(4, 0, 8, RPW_); } \
        else                      { if (npre == 1) GOR(3, 0, 1, RPW_); else GOR(3, 0, 4, RPW_); } } while (0)
    if (rpw == 8) GOP(8); else GOP(4);
#undef GOP
#undef GOR
    LAUNCH_CHECK();
    if (grid_out) *grid_out = grid;
    return CLLM_OK;
}
int launch_gemv_rows(hipStream_t st, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                     const float * bias, const float * resid) {
    if (epi != 0 && epi != 1) return CLLM_E_UNSUPPORTED;
    return gemv_rows_go(st, W, K, nrows, pro, px, pw, eps, epi, dst, bias, resid, nullptr);
}
// dst = W . quantize(RMS_NORM(px) * pw) AND the greedy sampler's first stage in the same launch: part_v[b] / part_i[b] = the largest of workgroup b's rows and its index
// (b < *np <= 256: the arrays k_argmax_final reduces).  CLLM_E_UNSUPPORTED: launch the mat-vec and k_argmax_partial separately.
int launch_gemv_rows_argmax(hipStream_t st, const void * W, int64_t K, int64_t nrows, const float * px, const float * pw, float eps, float * dst, float * part_v, int * part_i, int * np) {
    if (!part_v || !part_i || !np || nrows > INT32_MAX) return CLLM_E_UNSUPPORTED;
    return gemv_rows_go(st, W, K, nrows, 1, px, pw, eps, 2, dst, (const float *) part_v, (const float *) part_i, np);
}
