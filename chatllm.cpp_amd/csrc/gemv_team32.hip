// gemv_team32.hip -- the decode step's mat-vec for the 32-weight block formats (Q4_0 / Q4_1 / Q8_0) when there are FEW rows per CU (the hidden-sized outputs: qkv, o, down).
//
// Same contract and the same bits as k_gemv_dec / k_gemv_rows32 (activation produced in the kernel, dst = W . act (+ bias) (+ resid), every row accumulated in the order
// of the reference's AVX2 vec_dot: per block and AVX lane A, acc[A] = fma(d_w d_x, (float) sumi[A], acc[A])).  What is different is who runs the chains.
//   * k_gemv_dec gives a row to a wave: 64 blocks per step, then lanes 0..7 walk 64 records serially -- 64 fma + 32 LDS reads with 8 of 64 lanes at work, more
//     instructions than the integer part (160 wave instructions per 64 blocks).
//   * k_gemv_rows32 lets every lane run a chain (8 rows x 8 blocks per step, lane (r, j) = slot j of row r: 8 fma per step), but a wave then owns 8 rows: with 4096-8192
//     rows only 2-4 of a CU's 16 waves have work.
//   * here a TEAM of waves shares a unit of 8 rows: the EMIT waves split the unit's steps (8 rows x 8 blocks each) round robin, stream the weights and leave the
//     records of their steps in a double-buffered LDS slot; ONE chain wave per team consumes the steps in order, all 64 lanes busy (8 rows x 8 slots), and keeps the
//     accumulators in registers for the whole unit.  The hand-off is per ROUND (one step of every emit wave): the emit waves count their arrivals in an LDS word,
//     the chain wave waits for the count, walks the round's records with the LDS reads two steps ahead of the fmas, and publishes the number of rounds it has
//     consumed, which frees the slots of that parity.  All inside the workgroup, bounded waits (a timeout raises an error word instead of hanging).
// Teams per workgroup = units per CU (1, 2, 3 or 4: 16, 8, 5 or 4 waves per team); the teams' chain waves sit on different SIMDs.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"
#include "q32.h"

// weight steps in flight per emit wave (requested before the prologue, which lasts about as long as the whole matrix takes to stream)
// weight steps in flight per emit wave
#define T32_P_OF(FMT_, NPRE_) 3                       // (6 / 4 for Q4_0 / Q8_0, with or without T32_ACT_FIRST: the prologue barrier comes 0.6-1.6 us later, the launch 1.5-2 us -- measured)
#ifndef T32_ACT_FIRST
#define T32_ACT_FIRST 0                                // 1: the weight requests go out only once the wave's activation groups have landed (slower: measured)
#endif
typedef int t32_i4 __attribute__((ext_vector_type(4)));
#define T32_SLOT_BYTES (64 * 9 * 4 + 2 * 256)          // one step's records: [row r][slot j or d][t] floats (8 x 9 x 8), then Q4_1's m_w[64], s_a[64]
#define T32_SPINS (1 << 20)

__device__ __forceinline__ float t32_silu(float x) { return x / (1.0f + ggml_expf_poly(0.0f - x)); }
__device__ __forceinline__ float r32_silu_any(float x, bool body) { return body ? t32_silu(x) : x / (1.0f + libm_expf(-x)); }
__device__ __forceinline__ unsigned t32_lds_load(const unsigned * p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void t32_lds_store(unsigned * p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

struct t32_rec { f32x4 x0, x1, d0, d1, m0, m1, s0, s1; };

template <int FMT, int PRO, int NPRE>
__global__ void __launch_bounds__(1024) k_gemv_team32(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ W, int nblk, int nunits, float eps,
                                                      float * __restrict__ dst, const float * __restrict__ bias, const float * resid, int team /* waves per team: 4, 5, 8, 16 */,
                                                      unsigned * __restrict__ err, unsigned long long * ts) {
#define T32_TS(k) do { if (ts && (threadIdx.x & 63) == 0) ts[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 4 + (k)] = wall_clock64(); } while (0)
    T32_TS(0);
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ unsigned sy[20];                                        // [wave] the rounds this emit wave has handed over; [16 + team] the rounds the team's chain wave has consumed
    constexpr bool IS_Q8 = FMT == CLLM_TYPE_Q8_0, IS_41 = FMT == CLLM_TYPE_Q4_1;
    constexpr int BS = q32_fmt<FMT>::BS, P = T32_P_OF(FMT, NPRE);
    constexpr bool NEED_C0 = !IS_Q8 && !IS_41;                         // Q4_0's (nib - 8) . a = nib . a + c0: c0 = (-8, -8, -8, -8) . a per (block, AVX lane), a plane the prologue leaves behind the row
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = nblk * 32;
    const unsigned nb01 = (unsigned) nblk * (unsigned) BS;
    const unsigned arb = (unsigned) act_row_bytes(K, IS_41 ? ACT_Q8_1 : ACT_Q8_0);      // the activation row; Q4_0: then K bytes of c0[block][AVX lane] (int32)

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
    if (tid < 20) t32_lds_store(&sy[0] + tid, 0u);

    // ---- (2) roles.  Team q = wave / team (waves past the last whole team idle); the wave of the team that sits on SIMD q & 3 chains, the others emit.
    //          The team's units: u0 + ustride * k ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nteams = 16 / team, q = wave / team, wr = wave - q * team, E = team - 1;
    const int cr = (q * (65 - team)) & 3;                              // (q team + cr) & 3 == q & 3
    const bool in_team = q < nteams, is_emit = in_team && wr != cr;      // (the team's wave wr == cr chains)
    const int em = wr < cr ? wr : wr - 1;                              // emit index 0 .. E - 1
    const int S = (nblk + 7) >> 3;                                     // steps of 8 blocks per unit (S >= E: launcher); the blocks past a row's end in its last step are masked
    const int ustride = nteams * gridDim.x, u0 = q * gridDim.x + blockIdx.x;
    const int nmine = (in_team && u0 < nunits) ? (nunits - u0 + ustride - 1) / ustride : 0;
    const int total = nmine * S;                                       // the team's step sequence: g = k * S + s; round R = steps R E .. R E + E - 1, one per emit wave
    const int r = lane >> 3, t = lane & 7;
    const bool odd = !IS_41 && (t & 1);                                // rows are whole dwords: a block starts in the upper half of its first dword exactly when its index is odd
    const uint32_t psel = odd ? 0x07060504u : 0x05040302u;
    const unsigned lane_off = (unsigned) r * nb01 + (unsigned) t * BS - (odd ? 2u : 0u);
    // The kernel is bound by the number of instructions its waves issue (every kind: the SIMDs issue about one per four cycles across their waves -- profiles/r03_team32_*),
    // so the emit loop carries its cursors as running offsets (one add and a select per step) and has no control flow: waits are asm
    // blocks, trip counts are fixed up front -- with branches in the body the compiler rotates the ring of loaded registers through copies behind a full vmcnt(0).
    u32x4 qa[P], qb[IS_Q8 ? P : 1];
    uint32_t qt[P];
    const unsigned dA = (unsigned) E * 8u * BS, dB = (unsigned) ustride * 8u * nb01 - (unsigned)(S - E) * 8u * BS;      // to the wave's next step: inside the unit / into the team's next unit
    unsigned woff = (unsigned)(u0 < nunits ? u0 : 0) * (8u * nb01) + lane_off + (unsigned) em * 8u * BS;                // this lane's block of the wave's next requested step
    const unsigned wsafe = (unsigned)(u0 < nunits ? u0 : 0) * (8u * nb01) + lane_off;      // block t of the row's first step: inside the matrix for every lane (nblk >= 8)
    int is = em;                                                       // its step inside the unit
    int irem = (is_emit && em < total) ? (total - em + E - 1) / E : 0; // steps the wave has still to request (past the last: a block that exists, never used)
    auto issue = [&](int p) {
        // lanes whose block lies past the row's end in its last step (nblk % 8 != 0: masked by d = 0 at emit) must not READ there either: for the last row of
        // the matrix that is up to 7 blocks beyond the tensor -- they re-read a block that exists
        const unsigned o = (irem > 0 && is * 8 + t < nblk) ? woff : wsafe;
        irem--;
        const char * bp = W + o;                                       // (the matrix is smaller than 4 GiB: launcher)
        if (IS_41) { qt[p] = *(const uint32_t *) bp; qa[p] = *(const u32x4 *)(bp + 4); }
        else {
            qa[p] = *(const u32x4 *) bp;
            if (IS_Q8) { qb[IS_Q8 ? p : 0] = *(const u32x4 *)(bp + 16); qt[p] = *(const uint32_t *)(bp + 32); }
            else qt[p] = *(const uint32_t *)(bp + 16);
        }
        is += E;
        const bool wrap = is >= S;
        is = wrap ? is - S : is;
        woff += wrap ? dB : dA;
    };
    if (T32_ACT_FIRST) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (is_emit) {
#pragma unroll
        for (int p = 0; p < P; p++) issue(p);
    }

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
                v.x = r32_silu_any(p0.x, e + 0 < nv) * p0.y; v.y = r32_silu_any(p0.z, e + 1 < nv) * p0.w;
                v.z = r32_silu_any(p1.x, e + 2 < nv) * p1.y; v.w = r32_silu_any(p1.z, e + 3 < nv) * p1.w;
            }
            if (PRO == 4) {
                const f32x4 g = gg[u];
                v.x = r32_silu_any(v.x, e + 0 < nv) * g.x; v.y = r32_silu_any(v.y, e + 1 < nv) * g.y; v.z = r32_silu_any(v.z, e + 2 < nv) * g.z; v.w = r32_silu_any(v.w, e + 3 < nv) * g.w;
            }
            if (PRO == 1) { const f32x4 g = gg[u]; v.x = (v.x * scale) * g.x; v.y = (v.y * scale) * g.y; v.z = (v.z * scale) * g.z; v.w = (v.w * scale) * g.w; }
            quant4_store<32, IS_41>(lds, K, e, lane, v);
            if (NEED_C0) *(int *)(lds + arb + e) = dot4(0xf8f8f8f8u, *(const uint32_t *)(lds + e), 0);      // Q4_0: (nib - 8) . a = nib . a + c0, c0 = (-8, -8, -8, -8) . a per (block, AVX lane)
        }
    }
    __syncthreads();
    T32_TS(1);
    const unsigned long long cyc1 = ts ? clock64() : 0ull;
    if (nmine == 0 || !in_team) return;

    const char * act = lds;
    const float * actd = (const float *)(lds + act_off_d(K));
    const float * acts = (const float *)(lds + act_off_s(K, IS_41 ? ACT_Q8_1 : ACT_Q8_0));
    char * slots = lds + arb + (NEED_C0 ? K : 0);                      // [wave][2][T32_SLOT_BYTES]
    unsigned * done = &sy[16 + q];

    if (is_emit) {
        // ---- (3) emit: this wave's step of every round, records into its own two slots ----
        typedef __attribute__((address_space(3))) char * lds_p;
        const unsigned done_a = (unsigned)(size_t)(lds_p)(char *) done;
        unsigned flag_a = (unsigned)(size_t)(lds_p)(char *) &sy[wave];
        unsigned nspin = 0;                                            // all the wave's waits together are bounded: past the bound they fall through and the error word is raised
        int need = -1;                                                 // the slot of this parity is free once the chain wave has consumed round R - 2: done >= R - 1
        unsigned stamp = 1u;                                           // R + 1
        typedef __attribute__((address_space(3))) float * lds_f;
        const unsigned rec0 = (unsigned)(size_t)(lds_p)(slots + (size_t)(wave * 2) * T32_SLOT_BYTES) + (unsigned)(r * 72 + t) * 4u;
        unsigned rec = rec0;                                           // this lane's record column of the slot of round R (LDS address): X[slot * 8], X[64] = d_w d_x
        const unsigned rec_x = rec0 ^ (rec0 + (unsigned) T32_SLOT_BYTES), m_off = (unsigned)(64 * 9 + lane - (r * 72 + t)) * 4u;      // (to the other slot; to Q4_1's m_w[lane])
        int cs = em;                                                   // step of its unit of the wave's next step
        // the activation blocks of a step are read one step ahead (registers): the LDS latency stays off the step's dependency chain
        u32x4 a0n, a1n; t32_i4 c0n = {0, 0, 0, 0}, c1n = {0, 0, 0, 0}; float ydn, ysn = 0.0f; unsigned dnn = 0;
        const char * act_t = act + t * 32;
        const float * actd_t = actd + t, * acts_t = acts + t;
        auto act_fetch = [&](int s_next) {
            const char * ab = act_t + s_next * 256;
            a0n = *(const u32x4 *) ab; a1n = *(const u32x4 *)(ab + 16);
            ydn = actd_t[s_next * 8];
            if (IS_41) ysn = acts_t[s_next * 8];                       // (past the row's end: masked below together with m_w)
            if (NEED_C0) { c0n = *(const t32_i4 *)(ab + arb); c1n = *(const t32_i4 *)(ab + arb + 16); }
            dnn = t32_lds_load(done);
        };
        act_fetch(cs);
        auto step = [&](int p) {
            // (ordered behind the previous step's asm block: otherwise the scheduler lifts the first uses of all P ring entries to the top of the loop body -- a full wait again)
            if (IS_Q8) asm volatile("" : "+v"(qa[p]), "+v"(qb[IS_Q8 ? p : 0]), "+v"(qt[p]));
            else       asm volatile("" : "+v"(qa[p]), "+v"(qt[p]));
            uint32_t h; u32x4 q0, q1 = {0, 0, 0, 0};
            if (IS_41) { h = qt[p]; q0 = qa[p]; }
            else {
                const u32x4 a = qa[p];
                h = a.x >> (odd ? 16 : 0);
                auto pm = [&](uint32_t hi, uint32_t lo) { return __builtin_amdgcn_perm(hi, lo, psel); };
                if (IS_Q8) {
                    const u32x4 c = qb[IS_Q8 ? p : 0];
                    q0 = u32x4{pm(a.y, a.x), pm(a.z, a.y), pm(a.w, a.z), pm(c.x, a.w)};
                    q1 = u32x4{pm(c.y, c.x), pm(c.z, c.y), pm(c.w, c.z), pm(qt[p], c.w)};
                } else q0 = u32x4{pm(a.y, a.x), pm(a.z, a.y), pm(a.w, a.z), pm(qt[p], a.w)};
            }
            // (the request that refills ring entry p is ordered behind the last uses of its old contents: their live ranges must not overlap, or the ring is
            //  rotated through copies at the loop's end -- behind waits for every load in flight)
            if (IS_Q8) asm volatile("" : "+v"(woff) : "v"(q0), "v"(q1), "v"(h));
            else       asm volatile("" : "+v"(woff) : "v"(q0), "v"(h));
            issue(p);
            const u32x4 a0 = a0n, a1 = a1n; const t32_i4 c0 = c0n, c1 = c1n; const float yd = ydn, ys = ysn;
            unsigned dn = dnn;
            const bool ok = t < nblk - 8 * cs;
            cs += E; cs = cs >= S ? cs - S : cs;
            act_fetch(cs);
            int sm[8];
            if (IS_Q8) {
                sm[0] = dot4(q0.x, a0.x, 0); sm[1] = dot4(q0.y, a0.y, 0); sm[2] = dot4(q0.z, a0.z, 0); sm[3] = dot4(q0.w, a0.w, 0);
                sm[4] = dot4(q1.x, a1.x, 0); sm[5] = dot4(q1.y, a1.y, 0); sm[6] = dot4(q1.z, a1.z, 0); sm[7] = dot4(q1.w, a1.w, 0);
            } else {                                                   // Q4_0: c0 = the prologue's (-8 . a) plane, Q4_1: 0
                sm[0] = dot4(q0.x & 0x0f0f0f0fu, a0.x, c0.x); sm[1] = dot4(q0.y & 0x0f0f0f0fu, a0.y, c0.y);
                sm[2] = dot4(q0.z & 0x0f0f0f0fu, a0.z, c0.z); sm[3] = dot4(q0.w & 0x0f0f0f0fu, a0.w, c0.w);
                sm[4] = dot4((q0.x >> 4) & 0x0f0f0f0fu, a1.x, c1.x); sm[5] = dot4((q0.y >> 4) & 0x0f0f0f0fu, a1.y, c1.y);
                sm[6] = dot4((q0.z >> 4) & 0x0f0f0f0fu, a1.z, c1.z); sm[7] = dot4((q0.w >> 4) & 0x0f0f0f0fu, a1.w, c1.w);
            }
            // wait for done >= need (the value read a step ahead usually says so already)
            asm volatile("v_cmp_le_i32 vcc, %[need], %[dn]\n\t"
                         "s_cbranch_vccnz 2f\n"
                         "1:\n\t"
                         "s_sleep 1\n\t"
                         "ds_read_b32 %[dn], %[addr]\n\t"
                         "s_add_u32 %[n], %[n], 1\n\t"
                         "s_waitcnt lgkmcnt(0)\n\t"
                         "v_cmp_le_i32 vcc, %[need], %[dn]\n\t"
                         "s_cbranch_vccnz 2f\n\t"
                         "s_cmp_lt_u32 %[n], 0x100000\n\t"
                         "s_cbranch_scc1 1b\n"
                         "2:"
                         : [dn] "+v"(dn), [n] "+s"(nspin) : [need] "s"(need), [addr] "v"(done_a) : "vcc", "scc", "memory");
            lds_f X = (lds_f)(size_t) rec;                             // slot of AVX lane A = 4(A&1) + (A&2) + (A>>2): [A0 A4 A2 A6 | A1 A5 A3 A7] (the hsum below)
            X[0 * 8] = (float) sm[0]; X[1 * 8] = (float) sm[4]; X[2 * 8] = (float) sm[2]; X[3 * 8] = (float) sm[6];
            X[4 * 8] = (float) sm[1]; X[5 * 8] = (float) sm[5]; X[6 * 8] = (float) sm[3]; X[7 * 8] = (float) sm[7];
            X[8 * 8] = ok ? h2f((uint16_t) h) * yd : 0.0f;            // a block past the row's end: fma(0, finite, acc) = acc
            if (IS_41) { lds_f M = (lds_f)(size_t)(rec + m_off); M[0] = ok ? h2f((uint16_t)(h >> 16)) : 0.0f; M[64] = ok ? ys : 0.0f; }
            // the round goes to the chain wave: this wave's flag word = rounds handed over, written behind the records in the wave's LDS instruction stream
            // (the LDS executes a wave's instructions in order: no wait in between; every lane writes the same word)
            asm volatile("ds_write_b32 %0, %1" :: "v"(flag_a), "v"(stamp) : "memory");
            rec ^= rec_x;
            stamp += 1u; need += 1;
        };
        const int n_mine = em < total ? (total - em + E - 1) / E : 0, nfull = n_mine / P, ntail = n_mine - nfull * P;
        for (int gi = 0; gi < nfull; gi++) {                           // whole rotations of the prefetch ring
#pragma unroll
            for (int p = 0; p < P; p++) step(p);
        }
#pragma unroll
        for (int p = 0; p < P - 1; p++) if (p < ntail) step(p);        // (wave-uniform)
        if (nspin >= 0x100000u && lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        T32_TS(2);
        if (ts && lane == 0) ts[(blockIdx.x * 16 + wave) * 4 + 3] = clock64() - cyc1;
        return;
    }

    // ---- (4) chain: the team's rounds in order; lane (r, j) = slot j of row r.  Inside a round the records are read two steps ahead of the fmas ----
    __builtin_amdgcn_s_setprio(3);
    const int j = lane & 7;
    const int wbase = q * team;
    float acc = 0.0f, accs = 0.0f;
    int s_in_unit = 0, ku = 0;
    bool dead = false;
    // The LDS reads of the records are issued by hand (asm), three steps in flight, and waited for by count: left to the compiler the loop-carried reads are
    // waited for one step after their issue (its counter state at the loop header is the conservative merge), which exposes the LDS latency in every step.
    constexpr int NRD = IS_41 ? 8 : 4, DEPTH = IS_41 ? 2 : 3, WN = (DEPTH - 1) * NRD;   // reads per step, steps in flight (Q4_1: 32 registers a step); "at most WN outstanding" = the oldest step has landed
    const unsigned slots0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *) slots;
    const unsigned ox = (unsigned)(r * 72 + j * 8) * 4u, od = (unsigned)(r * 72 + 64) * 4u, om = (unsigned)(576 + r * 8) * 4u;
    auto fetch = [&](t32_rec & o, int e, int par) {
        const int w_abs = wbase + (e < cr ? e : e + 1);
        const unsigned base = slots0 + (unsigned)(w_abs * 2 + par) * (unsigned) T32_SLOT_BYTES;
        const unsigned ax = base + ox, ad = base + od;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %4 offset:16\n\tds_read_b128 %3, %5 offset:16"
                     : "=&v"(o.x0), "=&v"(o.d0), "=&v"(o.x1), "=&v"(o.d1) : "v"(ax), "v"(ad) : "memory");
        if (IS_41) {
            const unsigned am = base + om;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:256\n\tds_read_b128 %2, %4 offset:16\n\tds_read_b128 %3, %4 offset:272"
                         : "=&v"(o.m0), "=&v"(o.s0), "=&v"(o.m1), "=&v"(o.s1) : "v"(am) : "memory");
        }
    };
    auto landed = [&](t32_rec & o) {                                   // the uses of o below depend on this wait
        if (IS_41) asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(o.x0), "+v"(o.d0), "+v"(o.x1), "+v"(o.d1), "+v"(o.m0), "+v"(o.s0), "+v"(o.m1), "+v"(o.s1) : "n"(WN));
        else       asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o.x0), "+v"(o.d0), "+v"(o.x1), "+v"(o.d1) : "n"(WN));
    };
    // the unit's bias / residual values: requested a whole unit ahead by the lanes themselves (this wave has no other vector loads in flight to queue behind)
    float bv = 0.0f, rv = 0.0f;
    auto epi_fetch = [&](int k) {
        const int unit = k < nmine ? u0 + k * ustride : u0;
        if (bias)  bv = bias[(size_t) unit * 8 + r];
        if (resid) rv = resid[(size_t) unit * 8 + r];
    };
    auto chain = [&](const t32_rec & cur) {
        acc = __builtin_fmaf(cur.d0.x, cur.x0.x, acc); acc = __builtin_fmaf(cur.d0.y, cur.x0.y, acc); acc = __builtin_fmaf(cur.d0.z, cur.x0.z, acc); acc = __builtin_fmaf(cur.d0.w, cur.x0.w, acc);
        acc = __builtin_fmaf(cur.d1.x, cur.x1.x, acc); acc = __builtin_fmaf(cur.d1.y, cur.x1.y, acc); acc = __builtin_fmaf(cur.d1.z, cur.x1.z, acc); acc = __builtin_fmaf(cur.d1.w, cur.x1.w, acc);
        if (IS_41) {                                                   // summs = fma(m_w, s_a, summs) (arch/x86/quants.c:726): the same chain in every lane of the row
            accs = __builtin_fmaf(cur.m0.x, cur.s0.x, accs); accs = __builtin_fmaf(cur.m0.y, cur.s0.y, accs); accs = __builtin_fmaf(cur.m0.z, cur.s0.z, accs); accs = __builtin_fmaf(cur.m0.w, cur.s0.w, accs);
            accs = __builtin_fmaf(cur.m1.x, cur.s1.x, accs); accs = __builtin_fmaf(cur.m1.y, cur.s1.y, accs); accs = __builtin_fmaf(cur.m1.z, cur.s1.z, accs); accs = __builtin_fmaf(cur.m1.w, cur.s1.w, accs);
        }
        if (++s_in_unit == S) {                                        // 8 rows complete: hsum_float_8 over the 8 slots, epilogue, store
            float hsum = acc;
            hsum = hsum + dpp_f<DPP_QUAD_XOR1>(hsum); hsum = hsum + dpp_f<DPP_QUAD_XOR2>(hsum); hsum = hsum + dpp_f<DPP_HALF_MIRROR>(hsum);
            float v = IS_41 ? hsum + accs : hsum;
            const int unit = u0 + ku * ustride;
            if (bias)  v = v + bv;
            if (resid) v = v + rv;
            if (j == 0) dst[(size_t) unit * 8 + r] = v;
            acc = 0.0f; accs = 0.0f; s_in_unit = 0; ku++;
            epi_fetch(ku);
        }
    };
    const int nrounds = (total + E - 1) / E;
    const int fle = lane < E ? lane : E - 1;
    t32_rec A, B, Cc;
    epi_fetch(0);
    for (int R = 0, g0 = 0; R < nrounds; R++, g0 += E) {
        const int nR = total - g0 < E ? total - g0 : E, par = R & 1, last = nR - 1;
        if (!dead) {                                                   // every emit wave of the round has handed over (its flag word = rounds handed over >= R + 1): lane e looks at wave e's
            const unsigned * fl = &sy[wbase + (fle < cr ? fle : fle + 1)];
            int spins = 0;
            while (true) {
                const unsigned v = t32_lds_load(fl);
                if (!__builtin_amdgcn_ballot_w64(lane < nR && v < (unsigned)(R + 1))) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > T32_SPINS) { if (lane == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); dead = true; break; }
            }
        }
        if (DEPTH == 3) {
            fetch(A, 0, par); fetch(B, last < 1 ? last : 1, par);      // (past the round's end: the last step once more, never used)
            for (int e = 0; e < nR; e += 3) {
                fetch(Cc, e + 2 < last ? e + 2 : last, par); landed(A); chain(A);
                if (e + 1 < nR) { fetch(A, e + 3 < last ? e + 3 : last, par); landed(B); chain(B); }
                if (e + 2 < nR) { fetch(B, e + 4 < last ? e + 4 : last, par); landed(Cc); chain(Cc); }
            }
        } else {
            fetch(A, 0, par);
            for (int e = 0; e < nR; e += 2) {
                fetch(B, e + 1 < last ? e + 1 : last, par); landed(A); chain(A);
                if (e + 1 < nR) { fetch(A, e + 2 < last ? e + 2 : last, par); landed(B); chain(B); }
            }
        }
        asm volatile("" ::: "memory");                                 // behind the round's reads in this wave's LDS instruction stream: its slots go back to the emit waves
        if (lane == 0) t32_lds_store(done, (unsigned)(R + 1));
    }
    T32_TS(2);
    if (ts && lane == 0) ts[(blockIdx.x * 16 + wave) * 4 + 3] = clock64() - cyc1;
}

// the error word, per device: ONE page-locked host word mapped into the device (a timed-out wait stores its code there -- a store nobody else makes), so the check
// after a synchronize is a plain host read: no copy, no blocking call on the per-token path (the unmodified host's synchronize runs it too: cllm_check_kernel_errors)
static unsigned * g_t32_err_dev[CLLM_DEV_SLOTS];
static volatile unsigned * g_t32_err_host[CLLM_DEV_SLOTS];
static int g_team32_mode = -1;
static unsigned long long * g_t32_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_team32_ts(unsigned long long * dev_buf) { g_t32_ts = dev_buf; }   // tools only: [256 workgroups][16 waves][4] stamps
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_gemv_team32(int mode) { g_team32_mode = mode; }      // tests / tools: 0 off, 1 pick, 4 / 5 / 8 / 16 force the team size
extern "C" __attribute__((visibility("default"))) int cllm_debug_gemv_team32_error(void) {
    volatile unsigned * h = g_t32_err_host[dev_slot()];
    return h ? (int) *h : 0;
}

// after a synchronize: did a bounded in-kernel wait time out SINCE THE LAST CALL?  (the launch winds down instead of hanging; its results are void.)  The word is
// cleared when it is reported -- the caller has synchronized, nothing is running that could store to it -- so one timed-out hand-off fails one check, not every
// later one (include/chatllm_hip.h: cllm_check_kernel_errors)
int gemv_team32_check() {
    volatile unsigned * h = g_t32_err_host[dev_slot()];
    if (!h) return CLLM_OK;                                            // never launched on this device
    const unsigned e = *h;
    if (e) {
        *h = 0;
        if (e >= 300) FAIL(CLLM_E_HIP, "ffn_fused: the gather of a block of SiLU(gate) * up timed out (code %u: job %u): a workgroup of the launch never produced its features; set CLLM_FFN_FUSED=0", e, e - 300);
        FAIL(CLLM_E_HIP, "gemv_team32: a hand-off between the waves of a workgroup timed out (code %u); set CLLM_GEMV_TEAM32=0", e);
    }
    return CLLM_OK;
}
// the device's error word (shared by every kernel with bounded in-kernel waits): a page-locked host word mapped into the device
int kernel_error_word(unsigned ** dev_ptr) {
    unsigned * & g_t32_err = g_t32_err_dev[dev_slot()];
    if (!g_t32_err) {
        unsigned * h = nullptr;
        HIP_TRY(hipHostMalloc((void **) &h, 64, hipHostMallocMapped));
        *h = 0;
        HIP_TRY(hipHostGetDevicePointer((void **) &g_t32_err, h, 0));
        g_t32_err_host[dev_slot()] = h;
    }
    *dev_ptr = g_t32_err;
    return CLLM_OK;
}

// K % 32 == 0, rows whole dwords, nrows % 8 == 0, one unit (8 rows) per team; CLLM_E_UNSUPPORTED: the caller's other kernels take the launch
int launch_gemv_team32(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                       const float * bias, const float * resid) {
    if (g_team32_mode < 0) g_team32_mode = getenv("CLLM_GEMV_TEAM32") ? atoi(getenv("CLLM_GEMV_TEAM32")) : 1;
    const int mode = g_team32_mode;
    if (!mode || epi != 0 || (wtype != CLLM_TYPE_Q4_0 && wtype != CLLM_TYPE_Q4_1 && wtype != CLLM_TYPE_Q8_0)) return CLLM_E_UNSUPPORTED;
    const int bs = wtype == CLLM_TYPE_Q8_0 ? 34 : wtype == CLLM_TYPE_Q4_1 ? 20 : 18;
    if (K % 32 || ((K / 32) * bs) % 4 || pro < 1 || pro > 4 || nrows <= 0 || nrows % 8 || ((uintptr_t) W & 3) || (uint64_t) nrows * (uint64_t)(K / 32 * bs) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (K > ((pro == 2 || pro == 4) ? 32768 : 16384) || K < 256) return CLLM_E_UNSUPPORTED;
    const int nblk = (int)(K / 32), cus = device_cu_count();
    const int nunits = (int)(nrows / 8);
    // teams per workgroup = units per CU (each team takes ONE unit at a time; more than 4 units per CU: the other kernels have enough rows)
    int team = 0;
    if (mode == 4 || mode == 5 || mode == 8 || mode == 16) team = mode;
    else if (wtype != CLLM_TYPE_Q4_0 && K < 16384) team = 0;          // Q8_0 / Q4_1 at Llama-3-8B's shapes: faster per launch (down 21.8 -> 20.6 us) but not per decoded token (420 -> 417 tok/s in one call on one box): off
    else if (K < 8192 && nunits <= 2 * cus) team = 0;                 // short rows, one or two units per CU: a step or two per emit wave, the hand-offs are not amortized (k_gemv_dec is faster: measured)
    else if (nunits <= cus) team = 16;
    else if (nunits <= 2 * cus) team = 8;
    else if (nunits <= 3 * cus) team = 5;
    else if (nunits <= 4 * cus) team = 4;
    if (!team || (nblk + 7) / 8 < team - 1) return CLLM_E_UNSUPPORTED;      // (a unit has at least one step per emit wave)
    const int nteams = 16 / team;
    int grid = (nunits + nteams - 1) / nteams; if (grid > cus) grid = cus;
    const size_t lds = act_row_bytes(K, wtype == CLLM_TYPE_Q4_1 ? ACT_Q8_1 : ACT_Q8_0) + (wtype == CLLM_TYPE_Q4_0 ? (size_t) K : 0) + 32 * (size_t) T32_SLOT_BYTES;
    if (lds > 158 * 1024) return CLLM_E_UNSUPPORTED;
    unsigned * g_t32_err = nullptr;
    { const int erc = kernel_error_word(&g_t32_err); if (erc) return erc; }
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOT(FMT_, PRO_, NPRE_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_team32<FMT_, PRO_, NPRE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_team32<FMT_, PRO_, NPRE_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const char *) W, nblk, nunits, eps, dst, bias, resid, team, g_t32_err, g_t32_ts); } while (0)
#define GOP(FMT_) do { \
        if (pro == 1)      { if (npre == 1) GOT(FMT_, 1, 1); else GOT(FMT_, 1, 4); } \
        else if (pro == 2) { if (npre == 1) GOT(FMT_, 2, 1); else if (npre == 4) GOT(FMT_, 2, 4); else GOT(FMT_, 2, 8); } \
        else if (pro == 4) { if (npre == 1) GOT(FMT_, 4, 1); else if (npre == 4) GOT(FMT_, 4, 4); else GOT(FMT_, 4, 8); } \
        else               { if (npre == 1) GOT(FMT_, 3, 1); else GOT(FMT_, 3, 4); } } while (0)
    if (wtype == CLLM_TYPE_Q4_0) GOP(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOP(CLLM_TYPE_Q4_1); else GOP(CLLM_TYPE_Q8_0);
#undef GOP
#undef GOT
    LAUNCH_CHECK();
    return CLLM_OK;
}
