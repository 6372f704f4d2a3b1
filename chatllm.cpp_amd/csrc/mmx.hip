// mmx.hip -- quantized mat-mat for ANY number of activation columns in the reference's accumulation ORDER (bit-identical to libggml-cpu.so),
// the integer block dot products on the matrix cores.
//
// Replaces ggml_compute_forward_mul_mat (ggml-cpu/ggml-cpu.c:1229-1421) for prompts: tinyBLAS_Q0_AVX (llamafile/sgemm.cpp:1346-1790; Q4_0 / Q8_0 with
// >= 2 columns) and the vec_dot loop over ggml_vec_dot_q4_K_q8_K / q4_1_q8_1 (arch/x86/quants.c:1742-1822, 701-760).  All of them keep, per output element,
// 8 fp32 accumulators -- AVX lane A = the four elements 4A..4A+3 of every 32-element chunk -- and do ONE fma per block (Q4_K: per super-block) and lane in
// block order:  acc[A] = fma(d_w * d_x, (float) sumi[A], acc[A]);  result = hsum_float_8(acc) (+ the mins / summs chains).  The activation quantizers
// (discontinuous) downstream turn any other fp32 order into 0.1 sigma of logit noise (profiles/r02_prefill_modes.txt), so the order is the contract.
//
// sumi[A] is a FOUR-element integer dot product: exactly one K = 4 block of v_mfma_f32_16x16x4_4b_f16 (4 blocks = 4 AVX lanes per instruction, 16 x 16
// outputs each).  Operands are staged in LDS as integer-valued fp16 (weights: nib - 8 / nib / int8; Q4_K: sub-block scale * nib <= 945; activations:
// int8), products and sums are integers < 2^24: the fp32 result IS (float) sumi[A], no conversion.  The serial fp32 chains stay on the VALU as
// v_pk_fma_f32 (two chains per instruction) -- per 32-block and 16 x 16 patch: 2 MFMA (64 cycles) + 16 packed fma + 2 packed mul (72 cycles), the two
// pipes overlap across the two waves of a SIMD.  Q4_K folds once per super-block (sumi accumulates over the 8 sub-block groups inside the MFMA, C = D);
// its mins term m[2k] S[2k] + m[2k+1] S[2k+1] is a fifth K = 4 product with S split as 64 hi + lo (every operand exact in fp16).
//
// Tiling: 4 waves per workgroup, two workgroups per CU.  Q4_0 / Q4_1 / Q8_0: 64 weight rows (n) x 64 tokens (m), a wave owns 32 x 32 = 2 x 2 patches
// (128 accumulator registers); Q4_K (12 chains per output): 64 (n) x 32 (m), a wave owns 16 (n) x 32 (m).  K walked in steps of 128 (Q4_K: 256) elements:
// global -> registers one step ahead, registers -> LDS (unpack to fp16) between the two barriers of a step.
//   A operand = activations (rows = tokens), B operand = weights (cols = weight rows); D[v]: AVX lane 4 t + (v >> 2), token 4 (lane >> 4) + (v & 3), row lane & 15
//   (layouts checked on the device by tools/micro/mfma_probe.hip)
#include "common.h"
#include <stdio.h>
#include <stdlib.h>

typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct mmx_args {
    const char * W; int64_t nb01; int64_t N; int64_t K;
    const char * act; size_t act_stride; int64_t M;
    float * dst; int64_t ldd;              // dst[m * ldd + n]
    const float * resid; int64_t ldr;      // optional: dst = acc + resid[m * ldr + n] (the MUL_MAT -> ADD pair; resid may be dst itself)
    int epi;                               // 1: weight rows alternate gate_u, up_u; dst[m * ldd + u] = silu(acc[2u]) * acc[2u + 1]
};

template <int TYPE> struct mmx_traits { static constexpr int MI = 2, NJ = 2, WN = 2, WM = 2; };
template <> struct mmx_traits<CLLM_TYPE_Q4_K> { static constexpr int MI = 2, NJ = 1, WN = 4, WM = 1; };

// four bytes (unsigned, one per byte of u) -> fp16 pairs 1024 + byte: (elements 0, 1), (elements 2, 3).  0x64uu is the fp16 1024 + uu (ulp 1 in [1024, 2048))
__device__ __forceinline__ void bytes_to_h1024(uint32_t u, uint32_t & lo, uint32_t & hi) {
    lo = __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u);
    hi = __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u);
}
__device__ __forceinline__ uint32_t h2_sub(uint32_t x, uint32_t c) {          // per fp16 half: x - c (exact here)
    const h2v r = __builtin_bit_cast(h2v, x) - __builtin_bit_cast(h2v, c);
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t h2_fma(uint32_t x, uint32_t a, uint32_t c) {   // per fp16 half: x * a + c, one rounding (exact here)
    const h2v r = __builtin_elementwise_fma(__builtin_bit_cast(h2v, x), __builtin_bit_cast(h2v, a), __builtin_bit_cast(h2v, c));
    return __builtin_bit_cast(uint32_t, r);
}
// four int8 -> four fp16 of the same integers (two dwords)
__device__ __forceinline__ u32x2 i8x4_to_h(uint32_t x) {
    uint32_t lo, hi;
    bytes_to_h1024(x ^ 0x80808080u, lo, hi);                                  // byte + 128 as unsigned
    return u32x2{h2_sub(lo, 0x64806480u), h2_sub(hi, 0x64806480u)};           // - 1152
}
// 16 int8 -> 16 fp16
__device__ __forceinline__ void i8x16_to_h(const u32x4 q, u32x4 & o0, u32x4 & o1) {
    const u32x2 a = i8x4_to_h(q.x), b = i8x4_to_h(q.y), c = i8x4_to_h(q.z), d = i8x4_to_h(q.w);
    o0 = u32x4{a.x, a.y, b.x, b.y}; o1 = u32x4{c.x, c.y, d.x, d.y};
}
// four nibble bytes (0..15) -> four fp16 of nib - OFF (OFF = 8: Q4_0, 0: Q4_1)
template <int OFF> __device__ __forceinline__ u32x2 nib4_to_h(uint32_t nb) {
    uint32_t lo, hi;
    bytes_to_h1024(nb, lo, hi);
    constexpr uint32_t C = OFF == 8 ? 0x64086408u : 0x64006400u;              // 1032 / 1024
    return u32x2{h2_sub(lo, C), h2_sub(hi, C)};
}
// 16 nibble bytes -> 16 fp16
template <int OFF> __device__ __forceinline__ void nib16_to_h(uint32_t n0, uint32_t n1, uint32_t n2, uint32_t n3, u32x4 & o0, u32x4 & o1) {
    const u32x2 a = nib4_to_h<OFF>(n0), b = nib4_to_h<OFF>(n1), c = nib4_to_h<OFF>(n2), d = nib4_to_h<OFF>(n3);
    o0 = u32x4{a.x, a.y, b.x, b.y}; o1 = u32x4{c.x, c.y, d.x, d.y};
}
__device__ __forceinline__ uint32_t h2_splat(float v) { const _Float16 h = (_Float16) v; uint16_t b; __builtin_memcpy(&b, &h, 2); return (uint32_t) b * 0x00010001u; }
#ifdef MMX_SCALAR_FMA        // (A/B: two v_fma_f32 instead of one v_pk_fma_f32; build the file with -fno-slp-vectorize)
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
#else
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
#endif

template <int TYPE>
__global__ void __launch_bounds__(256, 2) k_mmx(const mmx_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    using T = mmx_traits<TYPE>;
    constexpr bool IS_K = TYPE == CLLM_TYPE_Q4_K, IS_41 = TYPE == CLLM_TYPE_Q4_1, IS_Q8 = TYPE == CLLM_TYPE_Q8_0;
    constexpr int MI = T::MI, NJ = T::NJ, BN = T::WN * NJ * 16, BM = T::WM * MI * 16, NT = 256;
    constexpr int KS = IS_K ? 256 : 128, NBK = KS / 32;             // elements of K per stage (the 32-block types stage 128: half the prefetch registers -- with 256 the kernel
                                                                    // spilled, and a scratch reload between the prefetch loads and the MFMA loop drains the prefetch)
    constexpr int LD = KS * 2 + (IS_K ? 32 : 0) + 16;               // bytes per tile row: KS fp16 (+ the mins pseudo-block) + pad (conflict-free ds_read_b64 fragments)
    constexpr int NPL = IS_41 ? 2 * NBK : IS_K ? 2 : NBK;           // f32 scale planes per weight row: dw[NBK] (+ mw[NBK]) / d, dmin
    constexpr int XPL = IS_41 ? 2 * NBK : IS_K ? 1 : NBK;           // per token: dx[NBK] (+ sx[NBK]) / dx
    constexpr int LDS_XT = 0, LDS_WT = BM * LD, LDS_WS = LDS_WT + BN * LD, LDS_XS = LDS_WS + BN * NPL * 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned mt, nt;                        // (L2-aware tile order: common.h gemm_tile_of)
    gemm_tile_of(blockIdx.x, (unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN), 512 / BM, mt, nt);
    const int64_t m0 = (int64_t) mt * BM, n0 = (int64_t) nt * BN;
    const int wn = (wave % T::WN) * (NJ * 16), wm = (wave / T::WN) * (MI * 16);
    const int l15 = lane & 15, l4 = lane >> 4;
    char * Xt = lds + LDS_XT; char * Wt = lds + LDS_WT; char * Ws = lds + LDS_WS; char * Xs = lds + LDS_XS;
    const int64_t K = a.K;
    const int64_t act_d = (int64_t) act_off_d(K), act_s = (int64_t) act_off_s(K, IS_K ? 256 : 32);

    // the chains: acc[patch][AVX lane][token pair]; Q4_K: + accm[patch][k]; Q4_1: + summs[patch]
    f32x2 acc[MI][NJ][8][2];
    f32x2 accm[IS_K ? MI : 1][IS_K ? NJ : 1][4][2];
    f32x2 summs[IS_41 ? MI : 1][IS_41 ? NJ : 1][2];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
#pragma unroll
            for (int A = 0; A < 8; A++) { acc[i][j][A][0] = f32x2{0.0f, 0.0f}; acc[i][j][A][1] = f32x2{0.0f, 0.0f}; }
            if (IS_K) {
#pragma unroll
                for (int k = 0; k < 4; k++) { accm[IS_K ? i : 0][IS_K ? j : 0][k][0] = f32x2{0.0f, 0.0f}; accm[IS_K ? i : 0][IS_K ? j : 0][k][1] = f32x2{0.0f, 0.0f}; }
            }
            if (IS_41) { summs[IS_41 ? i : 0][IS_41 ? j : 0][0] = f32x2{0.0f, 0.0f}; summs[IS_41 ? i : 0][IS_41 ? j : 0][1] = f32x2{0.0f, 0.0f}; }
        }

    // ---- staging: global -> registers (one K step ahead), registers -> LDS (unpack to fp16) ----
    constexpr int BS = IS_Q8 ? 34 : IS_41 ? 20 : 18, QOFF = IS_41 ? 4 : 2;
    struct __attribute__((packed, aligned(2))) q16 { uint32_t x, y, z, w; };
    constexpr int CPR = KS / 16;                                    // 16-byte chunks of int8 per token and stage
    constexpr int NXA = BM * CPR / NT;                              // activation tasks per thread: (token, 16 int8)
    constexpr int NXS = IS_K ? 1 : (BM * XPL + NT - 1) / NT;        // activation scale tasks (Q4_K: dx and the eight sub-block sums of token tid % BM ... see below)
    constexpr int NWT = IS_K ? BN * 4 / NT : BN * NBK / NT;         // weight tasks: Q4_K (row, 64-weight chunk) = 1; others (row, 32-block) = 1
    u32x4 rx[NXA]; uint32_t rxs[IS_K ? 3 : NXS];
    u32x4 rw[NWT], rw2[(IS_K || IS_Q8) ? NWT : 1], rwh[IS_K ? NWT : 1]; uint32_t rwd[IS_K ? 1 : NWT];      // rwd: the block's raw fp16 d (Q4_1: d | m << 16) -- converted at commit time:
    // a conversion here would make the prefetch wait for its own loads (s_waitcnt right behind them) and the global latency would sit in front of every K step
    auto prefetch = [&](int64_t k0) {
#pragma unroll
        for (int t = 0; t < NXA; t++) {
            const int c = tid + NT * t, row = c / CPR, ch = c % CPR;
            const int64_t m = m0 + row;
            rx[t] = u32x4{0, 0, 0, 0};
            if (m < a.M && k0 + ch * 16 < K) rx[t] = *(const u32x4 *)(a.act + m * a.act_stride + k0 + ch * 16);
        }
        if constexpr (IS_K) {          // thread (token = tid % BM, k = tid / BM) for tid < 4 BM: S[2k], S[2k+1]; k == 0 also dx
            rxs[0] = rxs[1] = rxs[2] = 0;
            if (tid < 4 * BM) {
                const int row = tid % BM, k = tid / BM;
                const int64_t m = m0 + row;
                if (m < a.M) {
                    const char * ar = a.act + m * a.act_stride;
                    rxs[0] = *(const uint32_t *)(ar + act_s + ((k0 / 32) + 2 * k) * 4);
                    rxs[1] = *(const uint32_t *)(ar + act_s + ((k0 / 32) + 2 * k + 1) * 4);
                    if (k == 0) rxs[2] = *(const uint32_t *)(ar + act_d + (k0 / 256) * 4);
                }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NXS; t++) {                         // dx[8][m] (Q4_1: then sx[8][m], the f32 s plane)
                const int c = tid + NT * t;
                rxs[t] = 0;
                if (c < BM * XPL) {
                    const int row = c % BM, pl = c / BM, sb = pl % NBK;
                    const int64_t m = m0 + row;
                    if (m < a.M && k0 + sb * 32 < K) rxs[t] = *(const uint32_t *)(a.act + m * a.act_stride + (pl < NBK ? act_d : act_s) + ((k0 / 32) + sb) * 4);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NWT; t++) {
            const int c = tid + NT * t;
            rw[t] = u32x4{0, 0, 0, 0};
            if constexpr (IS_K) {                                   // (row, chunk c4): header + 32 quant bytes
                const int row = c >> 2, c4 = c & 3;
                const int64_t n = n0 + row;
                rw2[t] = u32x4{0, 0, 0, 0}; rwh[t] = u32x4{0, 0, 0, 0};
                if (n < a.N) {
                    const char * bp = a.W + n * a.nb01 + (k0 / 256) * 144;
                    rwh[t] = *(const u32x4 *) bp;
                    rw[t]  = *(const u32x4 *)(bp + 16 + 32 * c4);
                    rw2[t] = *(const u32x4 *)(bp + 32 + 32 * c4);
                }
            } else {                                                // (row, 32-block sb)
                const int row = c / NBK, sb = c % NBK;
                const int64_t n = n0 + row, b = k0 / 32 + sb;
                rwd[IS_K ? 0 : t] = 0; if (IS_Q8) rw2[IS_Q8 ? t : 0] = u32x4{0, 0, 0, 0};
                if (n < a.N && b * 32 < K) {
                    const char * bp = a.W + n * a.nb01 + b * BS;
                    if (IS_41) rwd[IS_K ? 0 : t] = *(const uint32_t *) bp; else rwd[IS_K ? 0 : t] = *(const uint16_t *) bp;
                    const q16 q0 = *(const q16 *)(bp + QOFF);
                    rw[t] = u32x4{q0.x, q0.y, q0.z, q0.w};
                    if (IS_Q8) { const q16 q1 = *(const q16 *)(bp + 18); rw2[IS_Q8 ? t : 0] = u32x4{q1.x, q1.y, q1.z, q1.w}; }
                }
            }
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int t = 0; t < NXA; t++) {                             // 16 int8 -> 16 fp16
            const int c = tid + NT * t, row = c / CPR, ch = c % CPR;
            u32x4 o0, o1;
            i8x16_to_h(rx[t], o0, o1);
            *(u32x4 *)(Xt + row * LD + ch * 32)      = o0;
            *(u32x4 *)(Xt + row * LD + ch * 32 + 16) = o1;
        }
        if constexpr (IS_K) {
            if (tid < 4 * BM) {
                const int row = tid % BM, k = tid / BM;
                const int S0 = (int) rxs[0], S1 = (int) rxs[1];     // |S| <= 32 * 127: 64 hi + lo with lo in 0..63, hi in -64..63
                h4v v;
                v[0] = (_Float16)(float)(S0 & 63); v[1] = (_Float16)(float)(S0 >> 6); v[2] = (_Float16)(float)(S1 & 63); v[3] = (_Float16)(float)(S1 >> 6);
                *(h4v *)(Xt + row * LD + 512 + 8 * k) = v;
                if (k == 0) *(uint32_t *)(Xs + row * 4) = rxs[2];
            }
        } else {
#pragma unroll
            for (int t = 0; t < NXS; t++) {
                const int c = tid + NT * t;
                if (c < BM * XPL) *(uint32_t *)(Xs + c * 4) = rxs[t];          // plane-major: [pl][m]
            }
        }
#pragma unroll
        for (int t = 0; t < NWT; t++) {
            const int c = tid + NT * t;
            if constexpr (IS_K) {
                const int row = c >> 2, c4 = c & 3;
                const u32x4 h = rwh[t];
                const uint32_t u0 = h.y & 0x3f3f3f3fu, u2 = h.z & 0x3f3f3f3fu;
                const uint32_t u1 = (h.w & 0x0f0f0f0fu) | (((h.y >> 6) & 0x03030303u) << 4);
                const uint32_t u3 = ((h.w >> 4) & 0x0f0f0f0fu) | (((h.z >> 6) & 0x03030303u) << 4);
                const uint32_t scp = ((c4 & 2) ? u1 : u0) >> ((c4 & 1) * 16), mnp = ((c4 & 2) ? u3 : u2) >> ((c4 & 1) * 16);
                const float sc_lo = (float)(scp & 0xff), sc_hi = (float)((scp >> 8) & 0xff), m_lo = (float)(mnp & 0xff), m_hi = (float)((mnp >> 8) & 0xff);
                const uint32_t slo = h2_splat(sc_lo), shi = h2_splat(sc_hi);
                const uint32_t clo = h2_splat(-1024.0f * sc_lo), chi = h2_splat(-1024.0f * sc_hi);      // (1024 + nib) * sc - 1024 sc = nib * sc, one rounding, exact
                const uint32_t q[8] = { rw[t].x, rw[t].y, rw[t].z, rw[t].w, rw2[t].x, rw2[t].y, rw2[t].z, rw2[t].w };
                char * wr = Wt + row * LD + c4 * 128;               // elements 64 c4 .. : 32 low-nibble weights (sub-block 2 c4), then 32 high-nibble weights (2 c4 + 1)
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    uint32_t lo[8], hi[8];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const uint32_t x = q[4 * p + e];
                        uint32_t a0, a1, b0, b1;
                        bytes_to_h1024(x & 0x0f0f0f0fu, a0, a1); bytes_to_h1024((x >> 4) & 0x0f0f0f0fu, b0, b1);
                        lo[2 * e] = h2_fma(a0, slo, clo); lo[2 * e + 1] = h2_fma(a1, slo, clo); hi[2 * e] = h2_fma(b0, shi, chi); hi[2 * e + 1] = h2_fma(b1, shi, chi);
                    }
                    *(u32x4 *)(wr + p * 32)      = u32x4{lo[0], lo[1], lo[2], lo[3]}; *(u32x4 *)(wr + p * 32 + 16)      = u32x4{lo[4], lo[5], lo[6], lo[7]};
                    *(u32x4 *)(wr + 64 + p * 32) = u32x4{hi[0], hi[1], hi[2], hi[3]}; *(u32x4 *)(wr + 64 + p * 32 + 16) = u32x4{hi[4], hi[5], hi[6], hi[7]};
                }
                h4v mv;                                             // mins pseudo-block k = c4: [m2k, 64 m2k, m2k+1, 64 m2k+1]
                mv[0] = (_Float16) m_lo; mv[1] = (_Float16)(64.0f * m_lo); mv[2] = (_Float16) m_hi; mv[3] = (_Float16)(64.0f * m_hi);
                *(h4v *)(Wt + row * LD + 512 + 8 * c4) = mv;
                if (c4 == 0) { *(float *)(Ws + row * 4) = h2f((uint16_t)(h.x & 0xffff)); *(float *)(Ws + (BN + row) * 4) = h2f((uint16_t)(h.x >> 16)); }
            } else {
                const int row = c / NBK, sb = c % NBK;
                const u32x4 q0 = rw[t];
                u32x4 o[4];
                if constexpr (IS_Q8) {
                    i8x16_to_h(q0, o[0], o[1]); i8x16_to_h(rw2[IS_Q8 ? t : 0], o[2], o[3]);
                } else {
                    constexpr int OFF = IS_41 ? 0 : 8;              // elements 0..15 = low nibbles, 16..31 = high nibbles
                    nib16_to_h<OFF>(q0.x & 0x0f0f0f0fu, q0.y & 0x0f0f0f0fu, q0.z & 0x0f0f0f0fu, q0.w & 0x0f0f0f0fu, o[0], o[1]);
                    nib16_to_h<OFF>((q0.x >> 4) & 0x0f0f0f0fu, (q0.y >> 4) & 0x0f0f0f0fu, (q0.z >> 4) & 0x0f0f0f0fu, (q0.w >> 4) & 0x0f0f0f0fu, o[2], o[3]);
                }
                char * wr = Wt + row * LD + sb * 64;
                *(u32x4 *)(wr) = o[0]; *(u32x4 *)(wr + 16) = o[1]; *(u32x4 *)(wr + 32) = o[2]; *(u32x4 *)(wr + 48) = o[3];
                *(float *)(Ws + (sb * BN + row) * 4) = h2f((uint16_t)(rwd[IS_K ? 0 : t] & 0xffff));          // dw[8][n]
                if (IS_41) *(float *)(Ws + ((NBK + sb) * BN + row) * 4) = h2f((uint16_t)(rwd[IS_K ? 0 : t] >> 16));      // mw[NBK][n]
            }
        }
    };

    const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int frag_off = l15 * LD + l4 * 8;                         // row l15 of a patch, AVX lane l4 of the instruction's four
    prefetch(0);
    for (int64_t k0 = 0; k0 < K; k0 += KS) {
#ifdef MMX_NOSTAGE           // (timing experiments only: stage the first step, compute on it for every step)
        if (k0 == 0) { __syncthreads(); commit(); __syncthreads(); }
#else
        __syncthreads();                                            // the previous step's fragments are consumed
        commit();
        __syncthreads();
        if (k0 + KS < K) prefetch(k0 + KS);
#endif
#ifdef MMX_NOCOMPUTE
        continue;
#endif

        if constexpr (IS_K) {
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NJ; j++) {
                    const char * xr = Xt + (wm + i * 16) * LD + frag_off, * wr = Wt + (wn + j * 16) * LD + frag_off;
                    f32x16 D0 = zero16, D1 = zero16;
#pragma unroll
                    for (int g = 0; g < 8; g++) {                   // the 8 (chunk, nibble) groups of 32 elements: sumi accumulates inside the MFMA
                        const h4v x0 = *(const h4v *)(xr + g * 64), x1 = *(const h4v *)(xr + g * 64 + 32);
                        const h4v w0 = *(const h4v *)(wr + g * 64), w1 = *(const h4v *)(wr + g * 64 + 32);
                        D0 = __builtin_amdgcn_mfma_f32_16x16x4f16(x0, w0, D0, 0, 0, 0);
                        D1 = __builtin_amdgcn_mfma_f32_16x16x4f16(x1, w1, D1, 0, 0, 0);
                    }
                    const h4v xm = *(const h4v *)(xr + 512), wmn = *(const h4v *)(wr + 512);
                    const f32x16 Dm = __builtin_amdgcn_mfma_f32_16x16x4f16(xm, wmn, zero16, 0, 0, 0);
                    const float dw = *(const float *)(Ws + (wn + j * 16 + l15) * 4), dmw = *(const float *)(Ws + (BN + wn + j * 16 + l15) * 4);
                    const f32x4 dx = *(const f32x4 *)(Xs + (wm + i * 16 + l4 * 4) * 4);
                    // d = y.d * x.d, dmin = -y.d * x.dmin (arch/x86/quants.c:1766-1767; y = the activation)
                    const f32x2 dd0 = f32x2{dx.x * dw, dx.y * dw}, dd1 = f32x2{dx.z * dw, dx.w * dw};
                    const f32x2 dm0 = f32x2{(-dx.x) * dmw, (-dx.y) * dmw}, dm1 = f32x2{(-dx.z) * dmw, (-dx.w) * dmw};
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        acc[i][j][b][0]     = pk_fma(dd0, f32x2{D0[4 * b], D0[4 * b + 1]}, acc[i][j][b][0]);
                        acc[i][j][b][1]     = pk_fma(dd1, f32x2{D0[4 * b + 2], D0[4 * b + 3]}, acc[i][j][b][1]);
                        acc[i][j][4 + b][0] = pk_fma(dd0, f32x2{D1[4 * b], D1[4 * b + 1]}, acc[i][j][4 + b][0]);
                        acc[i][j][4 + b][1] = pk_fma(dd1, f32x2{D1[4 * b + 2], D1[4 * b + 3]}, acc[i][j][4 + b][1]);
                        accm[IS_K ? i : 0][IS_K ? j : 0][b][0] = pk_fma(dm0, f32x2{Dm[4 * b], Dm[4 * b + 1]}, accm[IS_K ? i : 0][IS_K ? j : 0][b][0]);
                        accm[IS_K ? i : 0][IS_K ? j : 0][b][1] = pk_fma(dm1, f32x2{Dm[4 * b + 2], Dm[4 * b + 3]}, accm[IS_K ? i : 0][IS_K ? j : 0][b][1]);
                    }
                }
        } else {
            const int nb = (int)((K - k0) / 32 < NBK ? (K - k0) / 32 : NBK);  // real 32-blocks of this step
#pragma unroll 1
            for (int s = 0; s < nb; s++) {
                h4v fx[MI][2], fw[NJ][2];
#pragma unroll
                for (int i = 0; i < MI; i++) { fx[i][0] = *(const h4v *)(Xt + (wm + i * 16) * LD + frag_off + s * 64); fx[i][1] = *(const h4v *)(Xt + (wm + i * 16) * LD + frag_off + s * 64 + 32); }
#pragma unroll
                for (int j = 0; j < NJ; j++) { fw[j][0] = *(const h4v *)(Wt + (wn + j * 16) * LD + frag_off + s * 64); fw[j][1] = *(const h4v *)(Wt + (wn + j * 16) * LD + frag_off + s * 64 + 32); }
                float dw[NJ]; f32x4 dx[MI];
#pragma unroll
                for (int j = 0; j < NJ; j++) dw[j] = *(const float *)(Ws + (s * BN + wn + j * 16 + l15) * 4);
#pragma unroll
                for (int i = 0; i < MI; i++) dx[i] = *(const f32x4 *)(Xs + (s * BM + wm + i * 16 + l4 * 4) * 4);
                // software pipeline over the MI x NJ patches of this block, ONE pair of result tiles: M0(0) M1(0) | F0(p) M0(p+1) F1(p) M1(p+1) ... -- every fold
                // (16 fma chains steps) runs under the MFMA issued just before it, and a tile is rewritten only after its fold (the compiler's own schedule
                // issued both MFMAs of a patch and then idled until the first result: the two pipes ran one after the other)
                constexpr int NP = MI * NJ;
                f32x16 D0 = __builtin_amdgcn_mfma_f32_16x16x4f16(fx[0][0], fw[0][0], zero16, 0, 0, 0);
                f32x16 D1 = __builtin_amdgcn_mfma_f32_16x16x4f16(fx[0][1], fw[0][1], zero16, 0, 0, 0);
#pragma unroll
                for (int p = 0; p < NP; p++) {
                    const int i = p / NJ, j = p % NJ, i2 = (p + 1) / NJ, j2 = (p + 1) % NJ;
                    // d = d_w * d_x (arch/x86/quants.c:556, 1021; sgemm.cpp tinyBLAS_Q0_AVX: unhalf(A.d) * unhalf(B.d))
                    const f32x2 dd0 = f32x2{dw[j] * dx[i].x, dw[j] * dx[i].y}, dd1 = f32x2{dw[j] * dx[i].z, dw[j] * dx[i].w};
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        acc[i][j][b][0] = pk_fma(dd0, f32x2{D0[4 * b], D0[4 * b + 1]}, acc[i][j][b][0]);
                        acc[i][j][b][1] = pk_fma(dd1, f32x2{D0[4 * b + 2], D0[4 * b + 3]}, acc[i][j][b][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < NP) D0 = __builtin_amdgcn_mfma_f32_16x16x4f16(fx[i2][0], fw[j2][0], zero16, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        acc[i][j][4 + b][0] = pk_fma(dd0, f32x2{D1[4 * b], D1[4 * b + 1]}, acc[i][j][4 + b][0]);
                        acc[i][j][4 + b][1] = pk_fma(dd1, f32x2{D1[4 * b + 2], D1[4 * b + 3]}, acc[i][j][4 + b][1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + 1 < NP) D1 = __builtin_amdgcn_mfma_f32_16x16x4f16(fx[i2][1], fw[j2][1], zero16, 0, 0, 0);
                }
                if constexpr (IS_41) {                              // summs = fma(m_w, s_x, summs) (arch/x86/quants.c:726; gcc contracts the statement)
                    float mw[NJ]; f32x4 sx[MI];
#pragma unroll
                    for (int j = 0; j < NJ; j++) mw[j] = *(const float *)(Ws + ((NBK + s) * BN + wn + j * 16 + l15) * 4);
#pragma unroll
                    for (int i = 0; i < MI; i++) sx[i] = *(const f32x4 *)(Xs + ((NBK + s) * BM + wm + i * 16 + l4 * 4) * 4);
#pragma unroll
                    for (int i = 0; i < MI; i++)
#pragma unroll
                        for (int j = 0; j < NJ; j++) {
                            summs[IS_41 ? i : 0][IS_41 ? j : 0][0] = pk_fma(f32x2{mw[j], mw[j]}, f32x2{sx[i].x, sx[i].y}, summs[IS_41 ? i : 0][IS_41 ? j : 0][0]);
                            summs[IS_41 ? i : 0][IS_41 ? j : 0][1] = pk_fma(f32x2{mw[j], mw[j]}, f32x2{sx[i].z, sx[i].w}, summs[IS_41 ? i : 0][IS_41 ? j : 0][1]);
                        }
                }
            }
        }
    }

    // ---- epilogue: hsum_float_8 (arch/x86/quants.c:43-49) (+ mins / summs), dst[m][n] ----
    const int64_t nv = (a.N / 2) & ~(int64_t) 7;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NJ; j++) {
            const int64_t n = n0 + wn + j * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                float x[8];
#pragma unroll
                for (int A = 0; A < 8; A++) x[A] = acc[i][j][A][r >> 1][r & 1];
                float r0 = x[4] + x[0], r1 = x[5] + x[1], r2 = x[6] + x[2], r3 = x[7] + x[3];
                r0 = r0 + r2; r1 = r1 + r3;
                float v = r0 + r1;
                if (IS_K) {
                    const float q0 = accm[IS_K ? i : 0][IS_K ? j : 0][0][r >> 1][r & 1], q1 = accm[IS_K ? i : 0][IS_K ? j : 0][1][r >> 1][r & 1];
                    const float q2 = accm[IS_K ? i : 0][IS_K ? j : 0][2][r >> 1][r & 1], q3 = accm[IS_K ? i : 0][IS_K ? j : 0][3][r >> 1][r & 1];
                    v = v + ((q0 + q2) + (q1 + q3));
                }
                if (IS_41) v = v + summs[IS_41 ? i : 0][IS_41 ? j : 0][r >> 1][r & 1];
                const int64_t m = m0 + wm + i * 16 + l4 * 4 + r;
                if (a.epi == 1) {           // the lane pair (2u, 2u + 1) holds gate_u and up_u of the same token: the even lane takes its neighbour's value and stores
                    const float up = dpp_f<DPP_QUAD_XOR1>(v);
                    const int64_t u = n >> 1;
                    if (!(lane & 1) && n + 1 < a.N && m < a.M) a.dst[m * a.ldd + u] = silu_any(v, u < nv) * up;
                } else if (n < a.N && m < a.M) a.dst[m * a.ldd + n] = a.resid ? v + a.resid[m * a.ldr + n] : v;
            }
        }
}

template <int TYPE> static constexpr int mmx_lds() {
    using T = mmx_traits<TYPE>;
    constexpr bool IS_K = TYPE == CLLM_TYPE_Q4_K, IS_41 = TYPE == CLLM_TYPE_Q4_1;
    constexpr int KS = IS_K ? 256 : 128, NBK = KS / 32;
    constexpr int BN = T::WN * T::NJ * 16, BM = T::WM * T::MI * 16, LD = KS * 2 + (IS_K ? 32 : 0) + 16;
    return (BM + BN) * LD + BN * (IS_41 ? 2 * NBK : IS_K ? 2 : NBK) * 4 + BM * (IS_41 ? 2 * NBK : IS_K ? 1 : NBK) * 4;
}

// the exact-order mat-mul: any M >= 1; K % 32 == 0 (Q4_K: % 256); CLLM_E_UNSUPPORTED: not this kernel's type
int launch_mmx(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, const tview & x, const tview & d, const float * resid, int64_t ldr, int epi) {
    if (d.nb[1] % 4) FAIL(CLLM_E_INVALID, "mmx: dst stride");
    mmx_args a;
    a.W = w.data; a.nb01 = w.nb[1]; a.N = w.ne[1]; a.K = w.ne[0];
    a.act = (const char *) act; a.act_stride = act_stride; a.M = x.ne[1];
    a.dst = (float *) d.data; a.ldd = d.nb[1] / 4; a.resid = resid; a.ldr = ldr; a.epi = epi;
    if (epi && (epi != 1 || resid || a.N % 2)) FAIL(CLLM_E_INVALID, "mmx: epilogue %d", epi);
#define GO(T) do { static uint64_t attr = 0; \
        using TR = mmx_traits<T>; constexpr int BN = TR::WN * TR::NJ * 16, BM = TR::WM * TR::MI * 16, LDS = mmx_lds<T>(); \
        if (((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN) > 0x7fffffff) FAIL(CLLM_E_UNSUPPORTED, "mmx: too many tiles"); \
        const dim3 grid((unsigned)(((a.M + BM - 1) / BM) * ((a.N + BN - 1) / BN))); \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_mmx<T>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); dev_flag_set(attr); } \
        hipLaunchKernelGGL(k_mmx<T>, grid, dim3(256), LDS, st, a); } while (0)
    static const bool dbg = getenv("CLLM_DEBUG") != nullptr;
    if (dbg) {
        int nb = 0;
        (void) hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void *) k_mmx<CLLM_TYPE_Q4_0>, 256, mmx_lds<CLLM_TYPE_Q4_0>());
        fprintf(stderr, "[cllm] mmx: occupancy query: %d workgroups of 256 threads per CU (Q4_0 form, %d bytes of LDS)\n", nb, mmx_lds<CLLM_TYPE_Q4_0>());
    }
    if (wtype == CLLM_TYPE_Q4_K) GO(CLLM_TYPE_Q4_K);
    else if (wtype == CLLM_TYPE_Q4_0) GO(CLLM_TYPE_Q4_0);
    else if (wtype == CLLM_TYPE_Q8_0) GO(CLLM_TYPE_Q8_0);
    else if (wtype == CLLM_TYPE_Q4_1) GO(CLLM_TYPE_Q4_1);
    else return CLLM_E_UNSUPPORTED;
#undef GO
    LAUNCH_CHECK();
    return CLLM_OK;
}
