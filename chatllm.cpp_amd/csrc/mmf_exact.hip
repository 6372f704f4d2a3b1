// mmf_exact.hip -- MUL_MAT with F16 src0 and MANY F32 src1 columns (the prompt's attention contractions K.Q and V.P) in the reference's
// accumulation ORDER, on the f32 matrix cores: bit-identical to libggml-cpu.so for every prompt length.
//
// Reference (ggml_compute_forward_mul_mat, ggml-cpu/ggml-cpu.c:1229-1421): src1 rows are rounded to fp16 (vec_dot_type of F16), then either
//   T8: tinyBLAS<8, __m256, __m256, ggml_fp16_t, ggml_fp16_t, float> (llamafile/sgemm.cpp:3835-3842, 477-640; taken for >= 2 columns, K % 8 == 0, rows % 4 == 0):
//       ONE 8-lane accumulator per output element, acc[l] = fma(w[8 s + l], x[8 s + l], acc[l]) over the steps s in order, hsum_float_8; or
//   VD: ggml_vec_dot_f16 (vec.cpp:264-, AVX2 + F16C): 32 accumulators, acc[a] = fma(w[32 s + a], x[32 s + a], acc[a]), GGML_F32x8_REDUCE's tree,
//       the K mod 32 leftovers added in double.
// v_mfma_f32_16x16x4_f32 IS a k-ordered fmaf chain: D = fma(a3, b3, fma(a2, b2, fma(a1, b1, fma(a0, b0, C)))), one rounding per step, bit for bit
// (tools/micro/mfma_probe.hip, profiles/r03_mfma_layout_probe.txt).  So accumulator a of a 16 x 16 patch of outputs is one MFMA accumulator tile,
// fed four chain steps per instruction: lane (i, g) supplies element 8 (4 u + g) + a (T8) / 32 (4 u + g) + a (VD) of row i.  The products of two
// fp16 values are exact in fp32, so the chain is the reference's.  64 FLOP / clk / SIMD = the packed-fp32 VALU peak, with the VALU left free.
//
// Tiles: operands staged in LDS as fp16 (src1 rounded with RNE on the way in), 128 elements of K per stage.  T8: workgroup 64 (rows) x 64 (columns),
// a wave owns 32 x 32 = 4 patches x 8 accumulators (128 registers); VD: workgroup 32 x 32, a wave owns one patch x 32 accumulators.
// Causal structure of the prompt (the runner / the module's attention pattern pass it): causal 1 (K.Q) skips tiles whose scores are all masked;
// causal 2 (V.P) ends the K loop where the probabilities of the tile's last column end -- fma(w, 0, acc) == acc, the skipped steps change no bit.
#include "common.h"
#include <type_traits>
#include <stdlib.h>

typedef _Float16 h8v __attribute__((ext_vector_type(8)));

struct mmfx_args {
    tview w, x, d;
    int causal, n_past;
    int x_f16;                  // src1 rows are already fp16 (the soft-max's OUT16 form): elements of 2 bytes, nothing to convert
    int zfirst;                 // grid order: 0: x = column tiles, z = the batch (heads); 1: x = the batch, z = column tiles (see mmfx_block)
};

// Causal products: a column tile's work grows with its index, and the hardware dispatches workgroups in x-then-y-then-z order.  With the heads on z, "longest first"
// held only inside a head: the last head's longest workgroups started when the rest of the chip was draining (K.Q by columns: 345 patches per wave slot on average, 256
// in the longest wave -- a tail of up to 74 %).  zfirst puts the HEADS on x, so every head's longest tiles are dispatched before anybody's short ones; the head index is
// permuted so that an XCD (workgroup id mod 8) keeps a contiguous range of heads -- the query heads of one K/V head share that XCD's L2.
__device__ __forceinline__ void mmfx_block(const mmfx_args & a, unsigned & bm, unsigned & gm, unsigned & bz) {
    if (a.zfirst) {
        gm = gridDim.z; bm = blockIdx.z;
        const unsigned Z = gridDim.x, bx = blockIdx.x;
        bz = (Z & 7) == 0 ? (bx & 7) * (Z >> 3) + (bx >> 3) : bx;
    } else { gm = gridDim.x; bm = blockIdx.x; bz = blockIdx.z; }
}

#define MMFX_KC 128
#define MMFX_LD (MMFX_KC * 2 + 16)

__device__ __forceinline__ u32x4 load8h(const char * p, int64_t e0, int64_t lim, bool al16) {      // elements e0..e0+7 of an fp16 row (p -> element e0), zero from `lim` on
    if (e0 + 8 <= lim && al16) return *(const u32x4 *) p;
    uint32_t r[4] = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) if (e0 + i < lim) r[i >> 1] |= (uint32_t) *(const uint16_t *)(p + 2 * i) << (16 * (i & 1));
    return u32x4{r[0], r[1], r[2], r[3]};
}
__device__ __forceinline__ u32x4 load8f_as_h(const char * p, int64_t e0, int64_t lim, bool al16) { // elements e0..e0+7 of an f32 row, rounded to fp16 (RNE), zero from `lim` on
    float v[8];
    if (e0 + 8 <= lim && al16) {
        const f32x4 a = *(const f32x4 *) p, b = *(const f32x4 *)(p + 16);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = e0 + i < lim ? *(const float *)(p + 4 * i) : 0.0f;
    }
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = (uint32_t) f2h(v[2 * i]) | ((uint32_t) f2h(v[2 * i + 1]) << 16);
    return u32x4{r[0], r[1], r[2], r[3]};
}
__device__ __forceinline__ void h8_to_f(const u32x4 r, float (&f)[8]) {
    const h8v h = __builtin_bit_cast(h8v, r);
#pragma unroll
    for (int i = 0; i < 8; i++) f[i] = (float) h[i];
}

// grid: x = column tiles (fastest), y = row tiles, z = src1's dims 2, 3
// PM: patches per wave along the columns (T8: 2 -> 64 x 64 workgroup tile, 128 accumulator registers, two workgroups per CU; 1 -> 64 rows x 32 columns, 64 accumulator
// registers, three workgroups per CU: the single-stage K.Q tiles are bound by their load / store latency, which more resident workgroups cover)
template <bool VD, int PM>
__global__ void __launch_bounds__(256, (VD || PM == 2) ? 2 : 3) k_mmf_exact(const mmfx_args a) {
    __shared__ __attribute__((aligned(16))) char lds[(VD ? 32 + 32 : 64 + 32 * PM) * MMFX_LD + (VD ? 2 * 32 * 64 : 0)];
    constexpr int BN = VD ? 32 : 64, BM = VD ? 32 : 32 * PM, PW = VD ? 1 : 2;       // PW (rows) x PM (columns) patches per wave
    constexpr int NACC = VD ? 32 : 8;
    const tview & w = a.w; const tview & x = a.x; const tview & d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    // causal 2 (V.P): a column tile's K loop ends at its last visible position, so the work grows with the tile index -- the LONGEST tiles are dispatched first
    // (the last tiles to start are then the shortest: the launch does not end on a few CUs walking 4096 positions)
    unsigned bm_, gm_, bz_; mmfx_block(a, bm_, gm_, bz_);
    const int64_t m0 = (int64_t)(a.causal == 2 ? gm_ - 1 - bm_ : bm_) * BM, n0 = (int64_t) blockIdx.y * BN;
    const int64_t i12 = bz_ % x.ne[2], i13 = bz_ / x.ne[2];
    const int64_t r2 = x.ne[2] / w.ne[2], r3 = x.ne[3] / w.ne[3];
    const int64_t K = w.ne[0], N = w.ne[1], M = x.ne[1];
    // causal 1: score (row n = position, column m = query) is masked when n > n_past + m: a tile whose first row lies beyond the last column's horizon is not computed
    if (a.causal == 1 && n0 > (int64_t) a.n_past + (m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1)) return;
    const char * wb = w.data + (i12 / r2) * w.nb[2] + (i13 / r3) * w.nb[3];
    const char * xb = x.data + i12 * x.nb[2] + i13 * x.nb[3];
    char * db = d.data + i12 * d.nb[2] + i13 * d.nb[3];
    const bool w_al = ((((uintptr_t) wb) | (uintptr_t) w.nb[1]) & 15) == 0, x_al = ((((uintptr_t) xb) | (uintptr_t) x.nb[1]) & 15) == 0;
    // the K range: VD's chains cover K & ~31 (the leftovers go through the double tail); causal 2 ends it at the tile's last visible position
    const int64_t Kmain = VD ? (K & ~(int64_t) 31) : K;
    int64_t kvis = K;                                                         // elements at or beyond kvis are zero in every column of this tile
    if (a.causal == 2) { const int64_t last = (m0 + BM - 1 < M - 1 ? m0 + BM - 1 : M - 1); kvis = (int64_t) a.n_past + last + 1 < K ? (int64_t) a.n_past + last + 1 : K; }
    const int64_t kend = Kmain < kvis ? Kmain : kvis;                         // main chains: [0, kend) rounded up to whole stages (zero-filled)
    char * Wt = lds; char * Xt = lds + BN * MMFX_LD;
    const int wn = (wave & 1) * (PW * 16), wm = (wave >> 1) * (PM * 16);

    f32x4 D[PM][PW][NACC];
#pragma unroll
    for (int i = 0; i < PM; i++)
#pragma unroll
        for (int j = 0; j < PW; j++)
#pragma unroll
            for (int q = 0; q < NACC; q++) D[i][j][q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    for (int64_t k0 = 0; k0 < kend; k0 += MMFX_KC) {
        __syncthreads();
        // ---- stage: BN weight rows and BM activation rows x 128 elements as fp16.  Every thread owns TW weight chunks and TX activation chunks of 8 elements;
        //      ALL their global loads are issued before the first is used (written as one loop with the conversion and the LDS store inside, the compiler kept
        //      one load in flight: eight memory round trips per stage, and the matrix cores idle for two thirds of the launch).  Whole, 16-byte aligned chunks
        //      take the wide path; a chunk cut by the row end / the causal horizon, or an unaligned one, goes element by element (load8h / load8f_as_h) ----
        {
            constexpr int TW = BN * 16 / 256, TX = BM * 16 / 256;
            u32x4 rw_[TW]; f32x4 xa_[TX], xb_[TX]; bool fw_[TW], fx_[TX];
#pragma unroll
            for (int t = 0; t < TW; t++) {
                const int c = tid + 256 * t, row = c >> 4, ch = c & 15;
                const int64_t e0 = k0 + ch * 8, n = n0 + row;
                fw_[t] = n < N && e0 + 8 <= kend && w_al;
                rw_[t] = u32x4{0, 0, 0, 0};
                if (fw_[t]) rw_[t] = *(const u32x4 *)(wb + n * w.nb[1] + e0 * 2);
            }
#pragma unroll
            for (int t = 0; t < TX; t++) {
                const int c = tid + 256 * t, row = c >> 4, ch = c & 15;
                const int64_t e0 = k0 + ch * 8, m = m0 + row;
                const int64_t lim = a.causal == 2 ? ((int64_t) a.n_past + m + 1 < kend ? (int64_t) a.n_past + m + 1 : kend) : kend;
                fx_[t] = m < M && e0 + 8 <= lim && x_al;
                xa_[t] = f32x4{0, 0, 0, 0}; xb_[t] = f32x4{0, 0, 0, 0};
                if (fx_[t]) {
                    if (a.x_f16) xa_[t] = *(const f32x4 *)(xb + m * x.nb[1] + e0 * 2);           // eight fp16, taken as they are
                    else { const char * p = xb + m * x.nb[1] + e0 * 4; xa_[t] = *(const f32x4 *) p; xb_[t] = *(const f32x4 *)(p + 16); }
                }
            }
#pragma unroll
            for (int t = 0; t < TW; t++) {
                const int c = tid + 256 * t, row = c >> 4, ch = c & 15;
                const int64_t e0 = k0 + ch * 8, n = n0 + row;
                u32x4 v = rw_[t];
                if (!fw_[t] && n < N && e0 < kend) v = load8h(wb + n * w.nb[1] + e0 * 2, e0, kend, false);
                *(u32x4 *)(Wt + row * MMFX_LD + ch * 16) = v;
            }
#pragma unroll
            for (int t = 0; t < TX; t++) {
                const int c = tid + 256 * t, row = c >> 4, ch = c & 15;
                const int64_t e0 = k0 + ch * 8, m = m0 + row;
                // causal 2: column m's probabilities end at n_past + m (what lies beyond was never written: read as zero)
                const int64_t lim = a.causal == 2 ? ((int64_t) a.n_past + m + 1 < kend ? (int64_t) a.n_past + m + 1 : kend) : kend;
                u32x4 v;
                if (fx_[t] && a.x_f16) v = __builtin_bit_cast(u32x4, xa_[t]);
                else if (fx_[t]) {
                    const f32x4 a4 = xa_[t], b4 = xb_[t];
                    v = u32x4{ (uint32_t) f2h(a4.x) | ((uint32_t) f2h(a4.y) << 16), (uint32_t) f2h(a4.z) | ((uint32_t) f2h(a4.w) << 16),
                               (uint32_t) f2h(b4.x) | ((uint32_t) f2h(b4.y) << 16), (uint32_t) f2h(b4.z) | ((uint32_t) f2h(b4.w) << 16) };
                } else {
                    v = u32x4{0, 0, 0, 0};
                    if (m < M && e0 < lim) v = a.x_f16 ? load8h(xb + m * x.nb[1] + e0 * 2, e0, lim, false) : load8f_as_h(xb + m * x.nb[1] + e0 * 4, e0, lim, false);
                }
                *(u32x4 *)(Xt + row * MMFX_LD + ch * 16) = v;
            }
        }
        __syncthreads();
        // ---- the chains: four steps per MFMA, lane group g supplies step 4 u + g ----
        if constexpr (!VD) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float xf[PM][8], wf[PW][8];
#pragma unroll
                for (int i = 0; i < PM; i++) h8_to_f(*(const u32x4 *)(Xt + (wm + i * 16 + l15) * MMFX_LD + u * 64 + g * 16), xf[i]);
#pragma unroll
                for (int j = 0; j < PW; j++) h8_to_f(*(const u32x4 *)(Wt + (wn + j * 16 + l15) * MMFX_LD + u * 64 + g * 16), wf[j]);
#pragma unroll
                for (int i = 0; i < PM; i++)
#pragma unroll
                    for (int j = 0; j < PW; j++)
#pragma unroll
                        for (int L = 0; L < 8; L++) D[i][j][L] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[i][L], wf[j][L], D[i][j][L], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; q++) {                                     // vector q of the four: accumulators 8 q + l
                float xf[8], wf[8];
                h8_to_f(*(const u32x4 *)(Xt + (wm + l15) * MMFX_LD + g * 64 + q * 16), xf);
                h8_to_f(*(const u32x4 *)(Wt + (wn + l15) * MMFX_LD + g * 64 + q * 16), wf);
#pragma unroll
                for (int L = 0; L < 8; L++) D[0][0][8 * q + L] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[L], wf[L], D[0][0][8 * q + L], 0, 0, 0);
            }
        }
    }

    // ---- VD: the K mod 32 leftovers of the tile's rows / columns -> LDS (fp16, zero-padded to 32) ----
    const int nleft = VD ? (int)(K - Kmain) : 0;
    char * Wl = lds + (BN + BM) * MMFX_LD; char * Xl = Wl + 32 * 64;
    if (VD && nleft > 0) {
        __syncthreads();
        for (int c = tid; c < 64 * 4; c += 256) {
            const int row = c >> 2, ch = c & 3;
            const int64_t e0 = Kmain + ch * 8;
            u32x4 v = u32x4{0, 0, 0, 0};
            if (row < 32) {
                const int64_t n = n0 + row;
                if (n < N && e0 < K) v = load8h(wb + n * w.nb[1] + e0 * 2, e0, K, false);
                *(u32x4 *)(Wl + row * 64 + ch * 16) = v;
            } else {
                const int64_t m = m0 + (row - 32);
                const int64_t lim = a.causal == 2 ? ((int64_t) a.n_past + m + 1 < K ? (int64_t) a.n_past + m + 1 : K) : K;
                if (m < M && e0 < lim) v = load8f_as_h(xb + m * x.nb[1] + e0 * 4, e0, lim, false);
                *(u32x4 *)(Xl + (row - 32) * 64 + ch * 16) = v;
            }
        }
        __syncthreads();
    }

    // ---- reduce + store: lane holds row n = .. + l15 and the four columns m = .. + 4 g + v ----
#pragma unroll
    for (int i = 0; i < PM; i++)
#pragma unroll
        for (int j = 0; j < PW; j++) {
            const int64_t n = n0 + wn + j * 16 + l15;
#pragma unroll
            for (int v = 0; v < 4; v++) {
                const int64_t m = m0 + wm + i * 16 + 4 * g + v;
                float res;
                if constexpr (!VD) {                                          // hsum_float_8
                    float s[8];
#pragma unroll
                    for (int L = 0; L < 8; L++) s[L] = D[i][j][L][v];
                    float r0 = s[4] + s[0], r1 = s[5] + s[1], r2_ = s[6] + s[2], r3_ = s[7] + s[3];
                    r0 = r0 + r2_; r1 = r1 + r3_;
                    res = r0 + r1;
                } else {                                                      // GGML_F32x8_REDUCE, then the leftovers in double
                    float s0[8];
#pragma unroll
                    for (int L = 0; L < 8; L++) { const float p = D[0][0][L][v] + D[0][0][16 + L][v], q = D[0][0][8 + L][v] + D[0][0][24 + L][v]; s0[L] = p + q; }
                    float t0[4];
#pragma unroll
                    for (int L = 0; L < 4; L++) t0[L] = s0[L] + s0[L + 4];
                    double sumf = (double)((t0[0] + t0[1]) + (t0[2] + t0[3]));
                    if (nleft > 0) {
                        const uint16_t * wl = (const uint16_t *)(Wl + (wn + l15) * 64), * xl = (const uint16_t *)(Xl + (wm + 4 * g + v) * 64);
                        for (int e = 0; e < nleft; e++) sumf += (double)(h2f(wl[e]) * h2f(xl[e]));
                    }
                    res = (float) sumf;
                }
                if (n < N && m < M) *(float *)(db + m * d.nb[1] + n * 4) = res;
            }
        }
}

// ---- the single-stage product (K <= 128: the prompt's K.Q, K = the head size) with NO LDS and NO barrier: the MFMA operand layout (lane = row l & 15, k group l >> 4:
//      eight consecutive K elements per chain step) is 16 contiguous bytes of an fp16 row, i.e. one global_load_dwordx4 per lane -- the operands go from global memory
//      straight into registers.  A WAVE owns 16 columns (queries): their Q values are converted once and held in 32 registers for the wave's life; it walks the K-cache
//      rows in patches of 16, three patches' loads in flight ahead of the one being multiplied (four register buffers); waves never wait for each other.  Measured at
//      cfg3 (32 heads, 4096 x 4096 causal, profiles/r04_prompt_attention_kq_forms.txt): 0.72 ms per launch against 1.00 (one workgroup per 64 x 32 tile) and 0.79 (a
//      workgroup walking the row tiles of its column tile through two LDS buffers, one barrier per tile), all three with the heads on grid x; the 34.1 M MFMAs alone are
//      0.46 ms, with the 32 conversions per 32 MFMAs 0.55 (tools/micro/mfma_f32_occupancy.hip: VALU instructions beside f32 MFMAs add their issue time).
//      Same products, same chains, same reduction: the same bits.
__global__ void __launch_bounds__(256, 3) k_mmf_exact_kq(const mmfx_args a) {
    const tview & w = a.w; const tview & x = a.x; const tview & d = a.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, g = lane >> 4;
    const int K = (int) w.ne[0], N = (int) w.ne[1], M = (int) x.ne[1];
    unsigned bm_, gm_, bz_; mmfx_block(a, bm_, gm_, bz_);
    const int m0 = ((int)(gm_ - 1 - bm_) * 4 + wave) * 16;                     // longest columns first (causal: the rows a column needs grow with it)
    if (m0 >= M) return;
    const unsigned z = bz_, ne12 = (unsigned) x.ne[2], r2 = (unsigned)(x.ne[2] / w.ne[2]), r3 = (unsigned)(x.ne[3] / w.ne[3]);
    const unsigned i12 = z % ne12, i13 = z / ne12;
    const char * wb = w.data + (int64_t)(i12 / r2) * w.nb[2] + (int64_t)(i13 / r3) * w.nb[3];
    const char * xb = x.data + (int64_t) i12 * x.nb[2] + (int64_t) i13 * x.nb[3];
    char * db = d.data + (int64_t) i12 * d.nb[2] + (int64_t) i13 * d.nb[3];
    const int kch = K >> 3;                                                    // 8-element chunks per row (the launcher: K % 8 == 0, K <= 128, rows 16-byte aligned)
    const int mlast = m0 + 15 < M - 1 ? m0 + 15 : M - 1;
    const int np_all = (N + 15) >> 4;
    // causal 1: score (row n, column m) is masked when n > n_past + m: patches whose first row lies beyond the last column's horizon are not computed
    const int np_vis = ((a.n_past + mlast) >> 4) + 1;
    const int npatch = a.causal == 1 && np_vis < np_all ? np_vis : np_all;

    // chain step u of lane group g is chunk 4 u + g of the row (elements 32 u + 8 g ..+7); chunks beyond K read a clamped address and count as zeros
    int coff[4]; bool cok[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int c = 4 * u + g; cok[u] = c < kch; coff[u] = (cok[u] ? c : kch - 1); }
    // this wave's Q values (column m0 + l15): f32 -> fp16 (RNE) -> f32, once
    float xf[4][8];
    {
        const int m = m0 + l15;
        const char * xp = xb + (int64_t)(m < M ? m : M - 1) * x.nb[1];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const f32x4 lo = *(const f32x4 *)(xp + coff[u] * 32), hi = *(const f32x4 *)(xp + coff[u] * 32 + 16);
            const bool ok = cok[u] && m < M;
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
            for (int i = 0; i < 8; i++) xf[u][i] = ok ? h2f(f2h(v[i])) : 0.0f;
        }
    }
    const char * wp[4];
#pragma unroll
    for (int u = 0; u < 4; u++) wp[u] = wb + coff[u] * 16;
    const int64_t wnb1 = w.nb[1];
    // this lane's four destination columns (m = m0 + 4 g + v) at row l15 of patch 0
    char * dp[4];
#pragma unroll
    for (int v = 0; v < 4; v++) { const int m = m0 + 4 * g + v; dp[v] = db + (int64_t)(m < M ? m : M - 1) * d.nb[1] + l15 * 4; }

    u32x4 rw[4][4];                                                            // [buffer][chain step]
    auto wload = [&](int pt, u32x4 (&r)[4]) {                                  // patch pt (clamped into the matrix) -> registers
        int n = pt * 16 + l15; n = n < N ? n : N - 1;
        const int64_t ro = (int64_t) n * wnb1;
#pragma unroll
        for (int u = 0; u < 4; u++) r[u] = *(const u32x4 *)(wp[u] + ro);
    };
    auto patch = [&](int pt, const u32x4 (&r)[4], auto full_tag) {
        constexpr bool FULL = decltype(full_tag)::value;
        f32x4 D[8];
#pragma unroll
        for (int L = 0; L < 8; L++) D[L] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const bool rok = FULL || pt * 16 + l15 < N;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            float wf[8];
            const uint32_t keep = (cok[u] && rok) ? 0xffffffffu : 0u;
            h8_to_f(u32x4{r[u].x & keep, r[u].y & keep, r[u].z & keep, r[u].w & keep}, wf);
#pragma unroll
            for (int L = 0; L < 8; L++) D[L] = __builtin_amdgcn_mfma_f32_16x16x4f32(xf[u][L], wf[L], D[L], 0, 0, 0);
        }
        const int n = pt * 16 + l15;
#pragma unroll
        for (int v = 0; v < 4; v++) {
            float r0 = D[4][v] + D[0][v], r1 = D[5][v] + D[1][v], r2_ = D[6][v] + D[2][v], r3_ = D[7][v] + D[3][v];      // hsum_float_8
            r0 = r0 + r2_; r1 = r1 + r3_;
            float * q = (float *)(dp[v] + (int64_t) pt * 64);
            if (FULL || (n < N && m0 + 4 * g + v < M)) *q = r0 + r1;
        }
    };
    // groups of four patches (one per register buffer); a group past the end repeats the last patch (same values stored again) -- no branch inside the loop body
    const int last = npatch - 1;
    wload(0, rw[0]); wload(1 < last ? 1 : last, rw[1]); wload(2 < last ? 2 : last, rw[2]);
    const int nfull = (m0 + 16 <= M) ? (N >> 4 < npatch ? N >> 4 : npatch) & ~3 : 0;       // leading groups wholly inside the matrix: stores without per-lane tests
    int pt = 0;
    for (; pt < nfull; pt += 4) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int pn = pt + s + 3 < last ? pt + s + 3 : last;
            wload(pn, rw[(s + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            patch(pt + s, rw[s], std::true_type{});
        }
    }
    for (; pt < npatch; pt += 4) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int pc = pt + s < last ? pt + s : last, pn = pt + s + 3 < last ? pt + s + 3 : last;
            wload(pn, rw[(s + 3) & 3]);
            __builtin_amdgcn_sched_barrier(0);
            patch(pc, rw[s], std::false_type{});
        }
    }
}

// w: F16 [K, N, ne02, ne03] (dense rows), x: F32 [K, M, ne12, ne13] (dense rows), d: F32 [N, M, ne12, ne13] (dense rows); CLLM_E_UNSUPPORTED: shapes this kernel does not take
int launch_mmf_exact(hipStream_t st, const tview & w, const tview & x, const tview & d, int causal, int n_past, bool x_f16) {
    if (w.nb[0] != 2 || x.nb[0] != (x_f16 ? 2 : 4) || d.nb[0] != 4 || (w.nb[1] | w.nb[2] | w.nb[3]) % 2 || (x.nb[1] | x.nb[2] | x.nb[3] | d.nb[1] | d.nb[2] | d.nb[3]) % 4) return CLLM_E_UNSUPPORTED;
    const int64_t K = w.ne[0], N = w.ne[1], M = x.ne[1], Z = x.ne[2] * x.ne[3];
    if (K <= 0 || N <= 0 || M <= 0 || Z <= 0 || Z > 65535) return CLLM_E_UNSUPPORTED;
    mmfx_args a; a.w = w; a.x = x; a.d = d; a.causal = causal; a.n_past = n_past; a.x_f16 = x_f16 ? 1 : 0;
    static const int zf_mode = getenv("CLLM_MMF_ZFIRST") ? atoi(getenv("CLLM_MMF_ZFIRST")) : 1;      // (tools: 0 = the heads on grid z as before)
    auto grid = [&](int64_t tm, int64_t tn) {                       // column tiles x row tiles x batch, in the order a.zfirst says
        a.zfirst = (zf_mode && causal != 0 && tm <= 65535) ? 1 : 0;
        return a.zfirst ? dim3((unsigned) Z, (unsigned) tn, (unsigned) tm) : dim3((unsigned) tm, (unsigned) tn, (unsigned) Z);
    };
    // llamafile_sgemm takes the product when n >= 2, k % 8 == 0, m % 4 == 0 (sgemm.cpp:3691, 488, 503-517); else the vec_dot loop
    const bool t8 = M >= 2 && K % 8 == 0 && N % 4 == 0;
    if (x_f16 && !t8) return CLLM_E_UNSUPPORTED;                    // (fp16 src1 rows: the tinyBLAS form only -- the caller's choice of path guarantees it)
    if (t8) {
        if ((N + 63) / 64 > 65535) return CLLM_E_UNSUPPORTED;
        // the narrow tile (64 rows x 32 columns, three workgroups per CU) for both contractions of the prompt's attention: measured at cfg3 (profiles/r04_mmf_exact_tile.txt)
        // 406.3 ms against 414.6-416.0 with the square tile and 413.5 with the narrow one on K.Q only -- a third resident workgroup covers the other two's stage loads
        static const int kq_mode = getenv("CLLM_MMF_KQ") ? atoi(getenv("CLLM_MMF_KQ")) : 1;       // (tools: 0 = the LDS-staged tiles for the single-stage shapes too)
        const bool al16 = ((((uintptr_t) w.data) | (uintptr_t) w.nb[1] | (uintptr_t) w.nb[2] | (uintptr_t) w.nb[3] | ((uintptr_t) x.data) | (uintptr_t) x.nb[1] | (uintptr_t) x.nb[2] | (uintptr_t) x.nb[3]) & 15) == 0;
        if (kq_mode && al16 && K <= MMFX_KC && !x_f16 && causal != 2 && M > 32 && N >= 256 && N < (1 << 30) && (M + 63) / 64 <= 65535) {      // a prompt's K.Q: waves walk the rows of their 16 columns
            const dim3 gr = grid((M + 63) / 64, 1);
            hipLaunchKernelGGL(k_mmf_exact_kq, gr, dim3(256), 0, st, a);
            LAUNCH_CHECK();
            return CLLM_OK;
        }
        static const int force_pm = getenv("CLLM_MMF_PM") ? atoi(getenv("CLLM_MMF_PM")) : 0;      // (tools: 2 forces the 64 x 64 tile)
        const int pm = force_pm == 2 ? 2 : 1;
        if (pm == 1) { const dim3 gr = grid((M + 31) / 32, (N + 63) / 64); hipLaunchKernelGGL((k_mmf_exact<false, 1>), gr, dim3(256), 0, st, a); }
        else         { const dim3 gr = grid((M + 63) / 64, (N + 63) / 64); hipLaunchKernelGGL((k_mmf_exact<false, 2>), gr, dim3(256), 0, st, a); }
    } else {
        if ((N + 31) / 32 > 65535) return CLLM_E_UNSUPPORTED;
        const dim3 gr = grid((M + 31) / 32, (N + 31) / 32);
        hipLaunchKernelGGL((k_mmf_exact<true, 1>), gr, dim3(256), 0, st, a);
    }
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ---- the prompt's eager attention block in the reference's order (calc_attn_scores / attn_scores_to_probs, src/layers.cpp:2541-2561, 2499-2539):
//      S = K.Q (scores never needed beyond the causal horizon are not computed), P = SOFT_MAX(DIAG_MASK_INF(SCALE(S))), ctx = V.P.
//      The score matrix [n_kv, qlen, heads] goes through a library-owned scratch buffer (stream_scratch, capi.hip), a group of heads at a time (<= 1 GiB).
int attn_prefill_exact(hipStream_t st, const tview & q, const tview & k, const tview & vt, char * dst, int64_t nbn, int64_t nbh, float scale, int n_past) {
    const int64_t hd = q.ne[0], qlen = q.ne[1], nh = q.ne[2], n_kv = k.ne[1], nkv = k.ne[2];
    if (nkv <= 0 || nh % nkv || k.ne[0] != hd || vt.ne[0] != n_kv || vt.ne[1] <= 0 || vt.ne[2] != nkv || q.ne[3] != 1 || k.ne[3] != 1 || n_kv != (int64_t) n_past + qlen) return CLLM_E_UNSUPPORTED;
    const int64_t r2 = nh / nkv;
    const size_t per_head = (size_t) n_kv * (size_t) qlen * 4;
    const size_t cap = (size_t) 1 << 30;
    int64_t hc = (int64_t)(cap / per_head) / r2 * r2;                 // heads per pass: whole GQA groups
    if (hc < r2) hc = r2;
    if (hc > nh) hc = nh;
    const size_t need = per_head * (size_t) hc;
    void * g_attn_scr = stream_scratch(st, SCRATCH_ATTN_SCORES, need);      // per (device, stream); an outgrown block stays alive for the launch lists captured with it
    if (!g_attn_scr) return CLLM_E_HIP;
    for (int64_t h0 = 0; h0 < nh; h0 += hc) {
        const int64_t nhc = h0 + hc <= nh ? hc : nh - h0;
        tview qv = q; qv.data += h0 * q.nb[2]; qv.ne[2] = nhc; qv.ne[3] = 1;
        tview kv = k; kv.data += (h0 / r2) * k.nb[2]; kv.ne[2] = nhc / r2; kv.ne[3] = 1;
        tview S; S.data = (char *) g_attn_scr; S.ne[0] = n_kv; S.ne[1] = qlen; S.ne[2] = nhc; S.ne[3] = 1;
        S.nb[0] = 4; S.nb[1] = n_kv * 4; S.nb[2] = n_kv * qlen * 4; S.nb[3] = S.nb[2] * nhc;
        int rc = launch_mmf_exact(st, kv, qv, S, 1, n_past);
        if (rc) return rc;
        // the probabilities as fp16 in place (what V.P's src1 conversion would make of them) where V.P runs in the tinyBLAS form; else f32 and the conversion while staging
        const bool vp_t8 = qlen >= 2 && n_kv % 8 == 0 && vt.ne[1] % 4 == 0;
        rc = vp_t8 ? launch_soft_max_causal_f16out(st, S, scale, n_past) : CLLM_E_UNSUPPORTED;
        const bool p16 = rc == CLLM_OK;
        if (rc == CLLM_E_UNSUPPORTED) {
            cllm_tensor St; St.type = CLLM_TYPE_F32; St.data = S.data;
            for (int i = 0; i < 4; i++) { St.ne[i] = S.ne[i]; St.nb[i] = (size_t) S.nb[i]; }
            rc = cllm_op_scale_mask_soft_max((void *) st, &St, &St, scale, n_past);
        }
        if (rc) return rc;
        tview vv = vt; vv.data += (h0 / r2) * vt.nb[2]; vv.ne[0] = n_kv; vv.ne[2] = nhc / r2; vv.ne[3] = 1;
        tview d; d.data = dst + h0 * nbh; d.ne[0] = vt.ne[1]; d.ne[1] = qlen; d.ne[2] = nhc; d.ne[3] = 1;
        d.nb[0] = 4; d.nb[1] = nbn; d.nb[2] = nbh; d.nb[3] = nbh * nhc;
        tview P = S; if (p16) P.nb[0] = 2;                            // (same rows, same strides: the fp16 values sit at the start of each row)
        rc = launch_mmf_exact(st, vv, P, d, 2, n_past, p16);
        if (rc) return rc;
    }
    return CLLM_OK;
}
