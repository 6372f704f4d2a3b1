// mmq.hip -- quantized mat-mat (prefill GEMM) on the CDNA4 int8 matrix cores.
//
// Replaces, for ne11 >= 9 columns, ggml_compute_forward_mul_mat (ggml-cpu.c:1229-1421) over the Q4_K vec_dot
// (ggml-cpu/quants.c:550-623) and, for Q4_0/Q8_0, llamafile's tinyBLAS_Q0 (ggml-cpu/llamafile/sgemm.cpp:1346-1790,
// 3910-3982).  SAME arithmetic as the CPU path: src1 is quantized to Q8_0 / Q8_K first, every
// 32-element block dot product is an exact int32 (here: one v_mfma_i32_16x16x32_i8 per 16x16 output patch),
// scaling and accumulation are fp32.  Only the fp32 summation ORDER differs from the CPU (tier T1).
//
// Tiling (per workgroup of 4 waves): 128 weight rows (n) x 64 tokens (m), K walked in steps of 256 elements
// (one Q4_K super-block / eight 32-blocks).  Per step the weight tile is unpacked to int8 in LDS ([n][256+16],
// the 16-byte pad makes the ds_read_b64 fragment reads conflict-free), the activation tile is copied from the
// act rows, block scales go to small [s][*] planes.  Wave w owns a 32(m) x 64(n) quadrant: 2x4 MFMA patches.
//   A operand = activations (rows = tokens), B operand = weights (cols = weight rows)
//   D[m][n]: lane holds n = lane&15 and the four tokens m = 4*(lane>>4) + r         (cdna_hip_programming.md section 3)
// Per 32-block and patch: 1 MFMA (16 x 16 x 32 MACs) + 4 VALU ops to fold the block scale in:
//   Q4_K : acc_i += sc[n][s] * D      (v_mad_i32_i24, exact)   ; per super-block: acc_f += (d[n]*dx[m]) * acc_i - (dmin[n]*dx[m]) * sum_s m[n][s]*sx[m][s]
//   Q4_0/Q8_0 : acc_f += float(D) * (dw[n][s] * dx[m][s])
// blockIdx.x walks the token tiles fastest so that concurrently resident workgroups share weight tiles in L2.
#include "common.h"

typedef int   i32x4 __attribute__((ext_vector_type(4)));

#define MMQ_BN 128
// token tile: MI 16-row patches per wave, two wave rows -> BM = 32 * MI tokens per workgroup.  Q4_0 / Q8_0 take 128 tokens (the
// weight unpack is amortised over twice the MACs); Q4_K carries int32 + fp32 accumulators and would spill: 64 tokens.
#define MMQ_LD 272           // bytes per LDS tile row: 256 int8 + 16 pad

template <int TYPE> struct mmq_traits;
// NW waves per workgroup, laid out 2 (n) x NW/2 (m); a wave owns 64 (n) x 16 MI (m).  Q4_0 / Q8_0 with EIGHT waves of 64 x 32 over the same 128 x 128 tile
// (four waves per SIMD, -DMMQ_NW_Q0=8) measured 136.0 ms against 132.4 ms for cfg3: the kernel is bound by the fold's VALU work AND the LDS fragment / scale
// reads together (9 KB per 32-block step and wave), and the smaller wave tile reads 1.5 x the fragments -- more waves do not help, so four it stays
#ifndef MMQ_NW_Q0
#define MMQ_NW_Q0 4
#endif
template <> struct mmq_traits<CLLM_TYPE_Q4_K> { static constexpr int kb = 256, MI = 2, NW = 4; };
template <> struct mmq_traits<CLLM_TYPE_Q4_0> { static constexpr int kb = 32,  MI = 16 / MMQ_NW_Q0, NW = MMQ_NW_Q0; };
template <> struct mmq_traits<CLLM_TYPE_Q8_0> { static constexpr int kb = 32,  MI = 16 / MMQ_NW_Q0, NW = MMQ_NW_Q0; };
template <> struct mmq_traits<CLLM_TYPE_Q4_1> { static constexpr int kb = 32,  MI = 2, NW = 4; };      // two more scale planes and operands per patch: 64 tokens

struct mmq_args {
    const char * W; int64_t nb01; int64_t N; int64_t K;
    const char * act; size_t act_stride; int64_t M;
    float * dst; int64_t ldd;     // dst[m * ldd + n]
    const float * resid; int64_t ldr;      // optional: dst = acc + resid[m * ldr + n] (the MUL_MAT -> ADD pair; resid may be dst itself)
    int epi;                               // 1: weight rows alternate gate_u, up_u; dst[m * ldd + u] = silu(acc[2u]) * acc[2u + 1]  (UNARY(SILU) -> MUL of BaseMLP::forward)
};

// LDS map (bytes)
//   Wt  : MMQ_BN * MMQ_LD                    int8 weights
//   Xt  : MMQ_BM * MMQ_LD                    int8 activations
//   Wsc : Q4_K: sc[n][8] u8 | mn[n][8] u8 | d[n] f32 | dmin[n] f32       others: dw[8][n] f32 [Q4_1: | mw[8][n] f32]
//   Xsc : Q4_K: dx[m] f32 | sx[8][m] i32                                   others: dx[8][m] f32 [Q4_1: | sx[8][m] f32]
constexpr int LDS_WT = 0;
constexpr int LDS_XT = LDS_WT + MMQ_BN * MMQ_LD;
constexpr int lds_ws(int bm) { return LDS_XT + bm * MMQ_LD; }
constexpr int lds_xs(int bm, bool q41) { return lds_ws(bm) + MMQ_BN * (q41 ? 16 : 8) * 4; }      // 4 KB (Q4_K uses 8+8+4+4 = 24 B per row), Q4_1: 8 KB
constexpr int lds_total(int bm, bool q41) { return lds_xs(bm, q41) + bm * (q41 ? 16 : 9) * 4; }   // two workgroups per CU must fit 160 KB

__device__ __forceinline__ uint32_t nib_minus8(uint32_t nib4) {   // four nibbles (one per byte, 0..15) -> four int8 (nib - 8)
    return (((nib4 | 0x80808080u) - 0x08080808u) ^ 0x80808080u);
}

template <int TYPE>
__global__ void __launch_bounds__(mmq_traits<TYPE>::NW * 64, mmq_traits<TYPE>::NW / 2) k_mmq(const mmq_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool IS_K = TYPE == CLLM_TYPE_Q4_K, IS_41 = TYPE == CLLM_TYPE_Q4_1;
    constexpr int MMQ_MI = mmq_traits<TYPE>::MI, NT = mmq_traits<TYPE>::NW * 64, MMQ_BM = (mmq_traits<TYPE>::NW / 2) * 16 * MMQ_MI;
    constexpr int LDS_WS = lds_ws(MMQ_BM), LDS_XS = lds_xs(MMQ_BM, TYPE == CLLM_TYPE_Q4_1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned mt, nt;                        // (L2-aware tile order: common.h gemm_tile_of)
    gemm_tile_of(blockIdx.x, (unsigned)((a.M + MMQ_BM - 1) / MMQ_BM), (unsigned)((a.N + MMQ_BN - 1) / MMQ_BN), 128 / MMQ_BM * 4, mt, nt);
    const int64_t m0 = (int64_t) mt * MMQ_BM, n0 = (int64_t) nt * MMQ_BN;
    const int wn = (wave & 1) * 64, wm = (wave >> 1) * (MMQ_MI * 16);       // this wave's quadrant inside the tile
    const int l15 = lane & 15, l4 = lane >> 4;

    char * Wt = lds + LDS_WT; char * Xt = lds + LDS_XT; char * Ws = lds + LDS_WS; char * Xs = lds + LDS_XS;
    const int64_t K = a.K;
    const int64_t act_d = (int64_t) act_off_d(K), act_s = (int64_t) act_off_s(K, mmq_traits<TYPE>::kb);      // (Q8_1 flavour: Q8_0 geometry)

    float acc_f[MMQ_MI][4][4];                                         // [m patch][n patch][r]
#pragma unroll
    for (int i = 0; i < MMQ_MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int r = 0; r < 4; r++) acc_f[i][j][r] = 0.0f;

    // ---- staging is split in two: global -> registers (issued one K step ahead, so the loads fly during the MFMA work of the
    //      current step) and registers -> LDS (unpack + store, between the two barriers of a step) ----
    constexpr int BS = TYPE == CLLM_TYPE_Q8_0 ? 34 : IS_41 ? 20 : 18, QOFF = IS_41 ? 4 : 2;      // block bytes, offset of the quants
    struct __attribute__((packed, aligned(2))) q16 { uint32_t x, y, z, w; };
    constexpr int NXA = MMQ_BM * 16 / NT;                          // activation chunk tasks per thread
    constexpr int NXS = IS_K ? (MMQ_BM * 9 + NT - 1) / NT : MMQ_BM * (IS_41 ? 16 : 8) / NT;      // activation scale tasks per thread
    constexpr int NWT = IS_K ? (MMQ_BN * 9 + NT - 1) / NT : MMQ_BN * 8 / NT;      // weight tasks per thread
    u32x4 rx[NXA]; uint32_t rxs[NXS]; u32x4 rw[NWT]; u32x4 rw2[IS_K ? 1 : NWT]; float rwd[IS_K ? 1 : NWT]; float rwm[IS_41 ? NWT : 1];
    auto prefetch = [&](int64_t k0) {
#pragma unroll
        for (int t = 0; t < NXA; t++) {                            // activations: MMQ_BM rows x 256 int8 (16 chunks of 16 B per row)
            const int c = tid + NT * t, row = c >> 4, ch = c & 15;
            const int64_t m = m0 + row;
            rx[t] = u32x4{0, 0, 0, 0};
            if (m < a.M && k0 + ch * 16 < K) rx[t] = *(const u32x4 *)(a.act + m * a.act_stride + k0 + ch * 16);
        }
#pragma unroll
        for (int t = 0; t < NXS; t++) {
            const int c = tid + NT * t;
            rxs[t] = 0;
            if (IS_K) {
                if (c < MMQ_BM * 9) {                              // dx[m], sx[8][m]
                    const int row = c % MMQ_BM, f = c / MMQ_BM;
                    const int64_t m = m0 + row;
                    if (m < a.M) {
                        const char * ar = a.act + m * a.act_stride;
                        rxs[t] = f == 0 ? *(const uint32_t *)(ar + act_d + (k0 / 256) * 4) : *(const uint32_t *)(ar + act_s + ((k0 / 32) + (f - 1)) * 4);
                    }
                }
            } else {                                               // dx[8][m] (Q4_1: then sx[8][m], the f32 s plane)
                const int row = c % MMQ_BM, pl = c / MMQ_BM, sb = pl & 7;
                const int64_t m = m0 + row;
                if (m < a.M && k0 + sb * 32 < K) rxs[t] = *(const uint32_t *)(a.act + m * a.act_stride + (pl < 8 ? act_d : act_s) + ((k0 / 32) + sb) * 4);
            }
        }
#pragma unroll
        for (int t = 0; t < NWT; t++) {
            const int c = tid + NT * t;
            rw[t] = u32x4{0, 0, 0, 0};
            if (IS_K) {
                if (c < MMQ_BN * 9) {                              // 9 chunks of 16 B per 144-byte super-block
                    const int row = c / 9, ch = c % 9;
                    const int64_t n = n0 + row;
                    if (n < a.N) rw[t] = *(const u32x4 *)(a.W + n * a.nb01 + (k0 / 256) * 144 + ch * 16);
                }
            } else {                                               // one 32-block per task
                const int row = c >> 3, sb = c & 7;
                const int64_t n = n0 + row, b = k0 / 32 + sb;
                rwd[IS_K ? 0 : t] = 0.0f; rw2[IS_K ? 0 : t] = u32x4{0, 0, 0, 0}; if (IS_41) rwm[IS_41 ? t : 0] = 0.0f;
                if (n < a.N && b * 32 < K) {
                    const char * bp = a.W + n * a.nb01 + b * BS;
                    rwd[IS_K ? 0 : t] = h2f(*(const uint16_t *) bp);
                    if (IS_41) rwm[IS_41 ? t : 0] = h2f(*(const uint16_t *)(bp + 2));
                    const q16 q0 = *(const q16 *)(bp + QOFF);
                    rw[t] = u32x4{q0.x, q0.y, q0.z, q0.w};
                    if (TYPE == CLLM_TYPE_Q8_0) { const q16 q1 = *(const q16 *)(bp + 18); rw2[IS_K ? 0 : t] = u32x4{q1.x, q1.y, q1.z, q1.w}; }
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int t = 0; t < NXA; t++) {
            const int c = tid + NT * t, row = c >> 4, ch = c & 15;
            *(u32x4 *)(Xt + row * MMQ_LD + ch * 16) = rx[t];
        }
#pragma unroll
        for (int t = 0; t < NXS; t++) {
            const int c = tid + NT * t;
            if (!IS_K || c < MMQ_BM * 9) *(uint32_t *)(Xs + ((c / MMQ_BM) * MMQ_BM + c % MMQ_BM) * 4) = rxs[t];
        }
#pragma unroll
        for (int t = 0; t < NWT; t++) {
            const int c = tid + NT * t;
            if (IS_K) {
                if (c < MMQ_BN * 9) {
                    const int row = c / 9, ch = c % 9;
                    const u32x4 v = rw[t];
                    if (ch == 0) {
                        const uint32_t u0 = v.y & 0x3f3f3f3fu, u2 = v.z & 0x3f3f3f3fu;
                        const uint32_t u1 = (v.w & 0x0f0f0f0fu) | (((v.y >> 6) & 0x03030303u) << 4);
                        const uint32_t u3 = ((v.w >> 4) & 0x0f0f0f0fu) | (((v.z >> 6) & 0x03030303u) << 4);
                        *(u32x2 *)(Ws + row * 8) = u32x2{u0, u1};                                   // sc[n][8]
                        *(u32x2 *)(Ws + MMQ_BN * 8 + row * 8) = u32x2{u2, u3};                      // mn[n][8]
                        *(float *)(Ws + MMQ_BN * 16 + row * 4) = h2f((uint16_t)(v.x & 0xffff));     // d[n]
                        *(float *)(Ws + MMQ_BN * 20 + row * 4) = h2f((uint16_t)(v.x >> 16));        // dmin[n]
                    } else {
                        const int j = ch - 1, off = 64 * (j >> 1) + 16 * (j & 1);
                        *(u32x4 *)(Wt + row * MMQ_LD + off)      = u32x4{v.x & 0x0f0f0f0fu, v.y & 0x0f0f0f0fu, v.z & 0x0f0f0f0fu, v.w & 0x0f0f0f0fu};
                        *(u32x4 *)(Wt + row * MMQ_LD + off + 32) = u32x4{(v.x >> 4) & 0x0f0f0f0fu, (v.y >> 4) & 0x0f0f0f0fu, (v.z >> 4) & 0x0f0f0f0fu, (v.w >> 4) & 0x0f0f0f0fu};
                    }
                }
            } else {
                const int row = c >> 3, sb = c & 7;
                const u32x4 q0 = rw[t];
                u32x4 lo, hi;
                if (TYPE == CLLM_TYPE_Q8_0) { lo = q0; hi = rw2[IS_K ? 0 : t]; }
                else if (IS_41) {                                  // nibbles stay 0..15: the minimum comes in through m_w * s_x
                    lo = u32x4{q0.x & 0x0f0f0f0fu, q0.y & 0x0f0f0f0fu, q0.z & 0x0f0f0f0fu, q0.w & 0x0f0f0f0fu};
                    hi = u32x4{(q0.x >> 4) & 0x0f0f0f0fu, (q0.y >> 4) & 0x0f0f0f0fu, (q0.z >> 4) & 0x0f0f0f0fu, (q0.w >> 4) & 0x0f0f0f0fu};
                } else {
                    lo = u32x4{nib_minus8(q0.x & 0x0f0f0f0fu), nib_minus8(q0.y & 0x0f0f0f0fu), nib_minus8(q0.z & 0x0f0f0f0fu), nib_minus8(q0.w & 0x0f0f0f0fu)};
                    hi = u32x4{nib_minus8((q0.x >> 4) & 0x0f0f0f0fu), nib_minus8((q0.y >> 4) & 0x0f0f0f0fu), nib_minus8((q0.z >> 4) & 0x0f0f0f0fu), nib_minus8((q0.w >> 4) & 0x0f0f0f0fu)};
                }
                *(u32x4 *)(Wt + row * MMQ_LD + sb * 32)      = lo;
                *(u32x4 *)(Wt + row * MMQ_LD + sb * 32 + 16) = hi;
                *(float *)(Ws + (sb * MMQ_BN + row) * 4) = rwd[IS_K ? 0 : t];       // dw[8][n]
                if (IS_41) *(float *)(Ws + ((8 + sb) * MMQ_BN + row) * 4) = rwm[IS_41 ? t : 0];      // mw[8][n]
            }
        }
    };

    prefetch(0);
    for (int64_t k0 = 0; k0 < K; k0 += 256) {
        __syncthreads();                                          // previous step's fragments are consumed
        commit();
        __syncthreads();
        if (k0 + 256 < K) prefetch(k0 + 256);

        // ---- compute ----
        int acc_i[MMQ_MI][4][4];
        if (IS_K) {
#pragma unroll
            for (int i = 0; i < MMQ_MI; i++)
#pragma unroll
                for (int j = 0; j < 4; j++)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc_i[i][j][r] = 0;
        }
        uint64_t scb[4];                                           // Q4_K: the 8 sub-block scales of this lane's weight row, per n patch
        if (IS_K) {
#pragma unroll
            for (int j = 0; j < 4; j++) scb[j] = *(const uint64_t *)(Ws + (wn + j * 16 + l15) * 8);
        }
#pragma unroll 1
        for (int s = 0; s < 8; s++) {
            long fa[MMQ_MI], fb[4];
#pragma unroll
            for (int i = 0; i < MMQ_MI; i++) fa[i] = *(const long *)(Xt + (wm + i * 16 + l15) * MMQ_LD + s * 32 + l4 * 8);
#pragma unroll
            for (int j = 0; j < 4; j++) fb[j] = *(const long *)(Wt + (wn + j * 16 + l15) * MMQ_LD + s * 32 + l4 * 8);
            // Q4_0 / Q4_1 / Q8_0: the 4 x MI patches of a 32-block are software-pipelined: the MFMA of patch p+1 is issued BEFORE the VALU work that folds
            // patch p's block scale in (sched_barrier pins that order), so the matrix core runs under the VALU instead of the wave
            // idling in s_nop until its result may be read (the compiler's own schedule: one D register set, 8 wait states per patch).
            constexpr int NP = 4 * MMQ_MI;
            if (IS_K) {             // (the same pipelining measured 2 % slower here: the kernel is at the VGPR limit already)
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int sc = (int)((scb[j] >> (8 * s)) & 0xff);
#pragma unroll
                    for (int i = 0; i < MMQ_MI; i++) {
                        const i32x4 d = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa[i], fb[j], i32x4{0, 0, 0, 0}, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) acc_i[i][j][r] = __mul24(sc, d[r]) + acc_i[i][j][r];
                    }
                }
            } else {
                float dw[4]; f32x4 dx[MMQ_MI];
#pragma unroll
                for (int j = 0; j < 4; j++) dw[j] = *(const float *)(Ws + (s * MMQ_BN + wn + j * 16 + l15) * 4);
#pragma unroll
                for (int i = 0; i < MMQ_MI; i++) dx[i] = *(const f32x4 *)(Xs + (s * MMQ_BM + wm + i * 16 + l4 * 4) * 4);
                // int32 -> f32 without v_cvt: the accumulator starts at the bit pattern of 1.5 * 2^23, so D's bits ARE the float
                // 12582912 + sum (exact: |sum| <= 32 * 127 * 127 < 2^22 keeps it inside the binade with ulp 1); one packed subtract
                // recovers float(sum) for two results at a time
                constexpr int MAGIC_I = 0x4B400000; constexpr float MAGIC_F = 12582912.0f;
                const i32x4 c0 = { MAGIC_I, MAGIC_I, MAGIC_I, MAGIC_I };
                i32x4 dcur = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa[0], fb[0], c0, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const int i = p % MMQ_MI, j = p / MMQ_MI;
                    i32x4 dnext = dcur;
                    if (p + 1 < NP) dnext = __builtin_amdgcn_mfma_i32_16x16x32_i8(fa[(p + 1) % MMQ_MI], fb[(p + 1) / MMQ_MI], c0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    const f32x4 df = __builtin_bit_cast(f32x4, dcur);      // (the whole vector: bit_cast of ONE element of an ext_vector reads element 0 with this compiler)
#pragma unroll
                    for (int r = 0; r < 4; r++) acc_f[i][j][r] = __builtin_fmaf(df[r] - MAGIC_F, dw[j] * dx[i][r], acc_f[i][j][r]);
                    __builtin_amdgcn_sched_barrier(0);
                    dcur = dnext;
                }
                if (IS_41) {                                       // + m_w[n][s] * s_x[m][s]  (ggml_vec_dot_q4_1_q8_1, quants.c:182)
                    float mw[4]; f32x4 sx[MMQ_MI];
#pragma unroll
                    for (int j = 0; j < 4; j++) mw[j] = *(const float *)(Ws + ((8 + s) * MMQ_BN + wn + j * 16 + l15) * 4);
#pragma unroll
                    for (int i = 0; i < MMQ_MI; i++) sx[i] = *(const f32x4 *)(Xs + ((8 + s) * MMQ_BM + wm + i * 16 + l4 * 4) * 4);
#pragma unroll
                    for (int j = 0; j < 4; j++)
#pragma unroll
                        for (int i = 0; i < MMQ_MI; i++)
#pragma unroll
                            for (int r = 0; r < 4; r++) acc_f[i][j][r] = __builtin_fmaf(mw[j], sx[i][r], acc_f[i][j][r]);
                }
            }
        }
        if (IS_K) {
            // per super-block float update, including the mins term  sum_s mn[n][s] * sx[m][s]  (exact int32)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int nl = wn + j * 16 + l15;
                const uint64_t mnb = *(const uint64_t *)(Ws + MMQ_BN * 8 + nl * 8);
                const float dn = *(const float *)(Ws + MMQ_BN * 16 + nl * 4), dmn = *(const float *)(Ws + MMQ_BN * 20 + nl * 4);
#pragma unroll
                for (int i = 0; i < MMQ_MI; i++) {
                    const int ml = wm + i * 16 + l4 * 4;
                    const f32x4 dx = *(const f32x4 *)(Xs + ml * 4);
                    int mins[4] = {0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < 8; s++) {
                        const i32x4 sx = *(const i32x4 *)(Xs + ((1 + s) * MMQ_BM + ml) * 4);
                        const int mn = (int)((mnb >> (8 * s)) & 0xff);
#pragma unroll
                        for (int r = 0; r < 4; r++) mins[r] = __mul24(mn, sx[r]) + mins[r];
                    }
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        acc_f[i][j][r] = __builtin_fmaf(dn * dx[r], (float) acc_i[i][j][r], acc_f[i][j][r]);
                        acc_f[i][j][r] = __builtin_fmaf(-(dmn * dx[r]), (float) mins[r], acc_f[i][j][r]);
                    }
                }
            }
        }
    }
    // ---- epilogue: dst[m][n] ----
    if (a.epi == 1) {                      // the lane pair (2u, 2u + 1) holds gate_u and up_u of the same tokens: the even lane takes its neighbour's value and stores
        const int64_t nv = (a.N / 2) & ~(int64_t) 7;               // ggml_vec_silu_f32: polynomial body below nv, libm tail (rows of a.N / 2 features)
#pragma unroll
        for (int i = 0; i < MMQ_MI; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int64_t n = n0 + wn + j * 16 + l15, u = n >> 1;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int64_t m = m0 + wm + i * 16 + l4 * 4 + r;
                    const float up = dpp_f<DPP_QUAD_XOR1>(acc_f[i][j][r]);
                    if (!(lane & 1) && n + 1 < a.N && m < a.M) a.dst[m * a.ldd + u] = silu_any(acc_f[i][j][r], u < nv) * up;
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < MMQ_MI; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int64_t n = n0 + wn + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int64_t m = m0 + wm + i * 16 + l4 * 4 + r;
                if (n < a.N && m < a.M) a.dst[m * a.ldd + n] = a.resid ? acc_f[i][j][r] + a.resid[m * a.ldr + n] : acc_f[i][j][r];
            }
        }
}

int launch_mmq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, const tview & x, const tview & d, const float * resid, int64_t ldr, int epi) {
    if (getenv("CLLM_NO_MMQ")) return CLLM_E_UNSUPPORTED;
    if (d.nb[1] % 4) FAIL(CLLM_E_INVALID, "mmq: dst stride");
    mmq_args a;
    a.W = w.data; a.nb01 = w.nb[1]; a.N = w.ne[1]; a.K = w.ne[0];
    a.act = (const char *) act; a.act_stride = act_stride; a.M = x.ne[1];
    a.dst = (float *) d.data; a.ldd = d.nb[1] / 4; a.resid = resid; a.ldr = ldr; a.epi = epi;
    if (epi && (epi != 1 || resid || a.N % 2)) FAIL(CLLM_E_INVALID, "mmq: epilogue %d", epi);

#define GO(T) do { static uint64_t attr = 0; \
        constexpr int BM = (mmq_traits<T>::NW / 2) * 16 * mmq_traits<T>::MI, LDS = lds_total(BM, T == CLLM_TYPE_Q4_1); \
        if (((a.M + BM - 1) / BM) * ((a.N + MMQ_BN - 1) / MMQ_BN) > 0x7fffffff) FAIL(CLLM_E_UNSUPPORTED, "mmq: too many tiles"); \
        const dim3 grid((unsigned)(((a.M + BM - 1) / BM) * ((a.N + MMQ_BN - 1) / MMQ_BN))); \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_mmq<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); dev_flag_set(attr); } \
        hipLaunchKernelGGL(k_mmq<T>, grid, dim3(mmq_traits<T>::NW * 64), LDS, st, a); } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GO(CLLM_TYPE_Q4_K);
    else if (wtype == CLLM_TYPE_Q4_0) GO(CLLM_TYPE_Q4_0);
    else if (wtype == CLLM_TYPE_Q8_0) GO(CLLM_TYPE_Q8_0);
    else if (wtype == CLLM_TYPE_Q4_1) GO(CLLM_TYPE_Q4_1);
    else return CLLM_E_UNSUPPORTED;
#undef GO
    LAUNCH_CHECK();
    return CLLM_OK;
}
