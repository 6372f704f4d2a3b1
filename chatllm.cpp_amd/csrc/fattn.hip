// fattn.hip -- GGML_OP_FLASH_ATTN_EXT and the same tiled kernel behind the prefill attention block: the score matrix never
// leaves the CU.
//
//   out[:, h, n] = sum_kv softmax_kv(scale * K[kv, h/r] . Q[n, h] + mask[kv, n]) * V[kv, h/r]
//   ggml_compute_forward_flash_attn_ext_f16 (ggml-cpu/ops.cpp:8114-8344 one query at a time, :8346-8640 tiled, :8642-8710 the
//   split-KV reduction); used by CoreAttention with `-fa` (src/layers.cpp:2634-2656) over the K / V caches of
//   src/layers.cpp:2925-2945 (F16, or Q8_0 with --cache_dtype q8_0).
//
// Numerics: TOLERANCE tier.  The CPU op itself has three summation orders (one_chunk / tiled / split-KV, chosen by shape and
// thread count), so there is no single bit pattern to match.  What is kept from the CPU path: Q is converted to K's
// vec_dot_type first (F16 K: rounded to fp16; Q8_0 K: quantize_row_q8_0, i.e. the 8-bit rounding of Q is reproduced), K / V
// are used at their stored precision, products are exact in fp32, the soft-max runs in fp32 (online, base 2).  What differs:
// P is rounded to fp16 before the P.V product (the CPU keeps it in fp32 but accumulates V in fp16 for an F16 cache), and
// accumulation is in MFMA order.
//
// One kernel, v_mfma_f32_32x32x16_f16, "swapped" form so that every soft-max quantity of a query row lives in one lane pair:
//   S^T[kv, q] = K[kv, :] . Q[q, :]      A = K tile (LDS, XOR-swizzled rows), B = Q (registers, loaded once)
//   lane (l31, hi) holds S^T[kv = crow(r, hi) + 32 j, q = l31], crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi
//   O^T[dv, q] += Vt[dv, kv] . P^T[kv, q]  A = V^T tile (LDS, [dv][kv] rows), B = P: exactly the registers the lane already
//   holds (the contraction order of an MFMA is free as long as A and B agree), so P never moves between lanes.
// A workgroup is NW waves x 32 query rows over one (kv head | head); K/V tiles of 64 positions are staged through LDS by all
// waves.  Decode (few rows): the r heads of a GQA group are packed as rows of one tile (R = r) so K/V are read once per
// group, NW = 1, and the positions are split over gridDim.x workgroups whose partial (m, l, O) are merged by k_fattn_merge.
#include "common.h"
#include <stdlib.h>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float    f32x16 __attribute__((ext_vector_type(16)));

#define FA_VSTR 136                  // bytes per V^T row in LDS: 64 positions * 2 B + 8 B (conflict-free ds_read_b64 over 32 rows)
#define FA_LOG2E 1.4426950408889634f
#ifndef FA_PD
#define FA_PD 4                      // LDS fragment reads are issued this many MFMAs ahead (2 and 8 measured the same: the loop is not latency-bound any more)
#endif
#define FA_TAU 8.0f                  // lazy soft-max reference: rescale only when a maximum grows by more than 2^8
#define FA_MAX_SPLITS 256            // split-KV decode: at most this many partials per (head, query)

struct fattn_args {
    const char * q; int64_t q_nb1, q_nb2, q_nb3;            // F32 [D, N, H, B]
    const char * k; int64_t k_nb1, k_nb2, k_nb3;            // [D, n_kv, Hkv, B] rows (F16 | Q8_0)
    const char * v; int64_t v_nb1, v_nb2, v_nb3;            // VL 0: [D, n_kv, Hkv, B] rows;  VL 1: [n_kv, D, Hkv, B] (F16, nb1 = bytes per dv row)
    const char * mask; int64_t m_nb1, m_nb2, m_nb3;         // F16 [n_kv, N, ne2, ne3]
    int m_ne2, m_ne3, m_al;                                 // m_al: rows are 8-byte aligned and n_kv % 4 == 0
    char * dst; int64_t d_nbn, d_nbh, d_nbb;                // F32 element (dv, h, n, b) at dst + n d_nbn + h d_nbh + b d_nbb + 4 dv
    float * part;                                           // splits > 1: [B][H][N][splits][D + 4] = O (unnormalized), m (log2 domain), l
    int N, H, Hkv, R, n_kv, n_past, splits, chunk;
    const int32_t * n_kv_dev;                               // decode inside a captured step: the number of cached positions is *n_kv_dev + 1 (n_kv: the upper bound)
    float sc2;                                              // scale * log2(e)
};

__device__ __forceinline__ int fa_crow(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// max over the two lanes (l31, 0) and (l31, 1) of a query row
__device__ __forceinline__ float fa_pair_max(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fa_pair_sum(float x) {
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// 8 consecutive elements e0 .. e0 + 7 (e0 % 8 == 0) of a Q8_0 row as fp16(d * q); blocks are 34 bytes, so 2-byte loads
__device__ __forceinline__ u32x4 fa_q8_8(const char * row, int e0) {
    const char * p = row + (e0 >> 5) * 34;
    const float d = h2f(*(const uint16_t *) p);
    const uint16_t * qs = (const uint16_t *)(p + 2 + (e0 & 31));
    u32x4 out;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint32_t w = qs[i];
        const _Float16 lo = (_Float16)(d * (float)(int8_t)(w & 0xff)), hi = (_Float16)(d * (float)(int8_t)(w >> 8));
        uint16_t a, b; __builtin_memcpy(&a, &lo, 2); __builtin_memcpy(&b, &hi, 2);
        out[i] = (uint32_t) a | ((uint32_t) b << 16);
    }
    return out;
}

template <int D, int KVT, int VL, int MASK, int NW>
__global__ void __launch_bounds__(NW * 64, 2) k_fattn(const fattn_args a) {
    constexpr int NT = NW * 64, BQ = NW * 32, KS = D / 16, NB = D / 32, CPR = D / 8;
    extern __shared__ __attribute__((aligned(16))) char fa_smem[];
    char * Kl = fa_smem, * Vl = fa_smem + 64 * D * 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int rows = a.N * a.R, nqb = (rows + BQ - 1) / BQ;
    int qb = (int) blockIdx.x % nqb;
    const int split = (int) blockIdx.x / nqb;
    if (MASK == 1) qb = nqb - 1 - qb;                                  // the long (late) query blocks first
    const int y = blockIdx.y, b = blockIdx.z;
    const int hk = y * a.R / (a.H / a.Hkv);
    const bool wave_live = qb * BQ + wave * 32 < rows;
    const int rho = qb * BQ + wave * 32 + l31;
    const bool rvalid = rho < rows;
    const int rc = rvalid ? rho : rows - 1;
    const int n = rc / a.R, h = y * a.R + rc % a.R;

    // ---- Q: 16 ks fragments of 8 halves (dk = 16 ks + 8 hi ..), converted as the CPU converts it for K's vec_dot
    half8 qf[KS];
    {
        const char * qp = a.q + (int64_t) n * a.q_nb1 + (int64_t) h * a.q_nb2 + (int64_t) b * a.q_nb3;
        float x[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ks++) {
            const f32x4 lo = *(const f32x4 *)(qp + (16 * ks + 8 * hi) * 4), up = *(const f32x4 *)(qp + (16 * ks + 8 * hi) * 4 + 16);
            x[ks][0] = lo.x; x[ks][1] = lo.y; x[ks][2] = lo.z; x[ks][3] = lo.w; x[ks][4] = up.x; x[ks][5] = up.y; x[ks][6] = up.z; x[ks][7] = up.w;
        }
        if (KVT == CLLM_TYPE_Q8_0) {                                   // quantize_row_q8_0 (arch/x86/quants.c:290-345) then the dequantized value
#pragma unroll
            for (int blk = 0; blk < D / 32; blk++) {
                float amax = 0.0f;
#pragma unroll
                for (int e = 0; e < 8; e++) amax = fmaxf(amax, fmaxf(fabsf(x[2 * blk][e]), fabsf(x[2 * blk + 1][e])));
                amax = fa_pair_max(amax);
                const float d = amax / 127.0f, id = amax != 0.0f ? 127.0f / amax : 0.0f, dq = h2f(f2h(d));
#pragma unroll
                for (int e = 0; e < 8; e++) { x[2 * blk][e] = dq * rintf(x[2 * blk][e] * id); x[2 * blk + 1][e] = dq * rintf(x[2 * blk + 1][e] * id); }
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ks++)
#pragma unroll
            for (int e = 0; e < 8; e++) qf[ks][e] = (_Float16) x[ks][e];
    }

    const char * kb = a.k + (int64_t) hk * a.k_nb2 + (int64_t) b * a.k_nb3;
    const char * vb = a.v + (int64_t) hk * a.v_nb2 + (int64_t) b * a.v_nb3;
    const char * mrow = MASK == 2 ? a.mask + (int64_t) n * a.m_nb1 + (int64_t)(h % a.m_ne2) * a.m_nb2 + (int64_t)(b % a.m_ne3) * a.m_nb3 : nullptr;

    const int n_kv = a.n_kv_dev ? min(a.n_kv, a.n_kv_dev[0] + 1) : a.n_kv;
    const int kv_lo = split * a.chunk;
    int kv_hi = min(n_kv, kv_lo + a.chunk);
    if (MASK == 1) kv_hi = min(kv_hi, a.n_past + min(a.N - 1, (qb * BQ + BQ - 1) / a.R) + 1);
    const int lim = MASK == 1 ? a.n_past + n : 0;                      // causal: position kv is visible to row n iff kv <= n_past + n

    const float mask_mul = FA_LOG2E / a.sc2;                           // mask values are added to the RAW scores: (s + mask log2e / sc2) sc2 = s sc2 + mask log2e
    f32x16 o[NB];
#pragma unroll
    for (int i = 0; i < NB; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) o[i][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;

    // ---- K / V staging is split in two: global -> registers (issued one tile ahead: the loads fly under the MFMA work of the current tile) and
    //      registers -> LDS (conversion, swizzle, transposition; between the two barriers of a tile).  Q8_0 rows keep their raw 16-bit pieces in
    //      the registers (34-byte blocks: 2-byte alignment) and are dequantized on the way into LDS.
    constexpr bool Q8 = KVT == CLLM_TYPE_Q8_0;
    constexpr int NKC = (64 * CPR + NT - 1) / NT;                     // K chunks (8 elements) per thread
    constexpr int NVU = VL == 1 ? (D * 8 + NT - 1) / NT : (16 * CPR + NT - 1) / NT;      // V^T chunks (VL 1) / 4-row x 8-element units (VL 0) per thread
    struct raw8 { uint32_t h[4]; uint32_t d; };                      // Q8_0: four 16-bit pieces of 8 quants + the block scale
    u32x4 kreg[Q8 ? 1 : NKC]; raw8 kraw[Q8 ? NKC : 1];
    u32x4 vreg[Q8 ? 1 : (VL == 1 ? NVU : 4 * NVU)]; raw8 vraw[Q8 ? 4 * NVU : 1];
    auto q8_load = [&](const char * row, int e0) {
        raw8 r; const char * p = row + (e0 >> 5) * 34;
        r.d = *(const uint16_t *) p;
        const uint16_t * qs = (const uint16_t *)(p + 2 + (e0 & 31));
#pragma unroll
        for (int i = 0; i < 4; i++) r.h[i] = qs[i];
        return r;
    };
    auto q8_cvt = [&](const raw8 & r) {
        const float d = h2f((uint16_t) r.d);
        u32x4 out;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const _Float16 lo = (_Float16)(d * (float)(int8_t)(r.h[i] & 0xff)), up = (_Float16)(d * (float)(int8_t)(r.h[i] >> 8));
            uint16_t x, y; __builtin_memcpy(&x, &lo, 2); __builtin_memcpy(&y, &up, 2);
            out[i] = (uint32_t) x | ((uint32_t) y << 16);
        }
        return out;
    };
    auto load_tile = [&](int kv0) {
#pragma unroll
        for (int t = 0; t < NKC; t++) {
            const int c = tid + t * NT, row = c / CPR, col = c % CPR, kv = min(kv0 + (c < 64 * CPR ? row : 0), kv_hi - 1);
            if (Q8) kraw[Q8 ? t : 0] = q8_load(kb + (int64_t) kv * a.k_nb1, col * 8);
            else    kreg[Q8 ? 0 : t] = *(const u32x4 *)(kb + (int64_t) kv * a.k_nb1 + col * 16);
        }
        if (VL == 1) {
#pragma unroll
            for (int t = 0; t < NVU; t++) {
                const int c = tid + t * NT, dv = (c >> 3) % D, kc = c & 7, kvs = kv0 + 8 * kc;
                const int kvl = min(kvs, (n_kv - 1) & ~7);
                vreg[Q8 ? 0 : t] = *(const u32x4 *)(vb + (int64_t) dv * a.v_nb1 + (int64_t) kvl * 2);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NVU; t++) {
                const int u = (tid + t * NT) % (16 * CPR), kvg = u / CPR, dvg = u % CPR;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int kv = min(kv0 + 4 * kvg + i, kv_hi - 1);
                    if (Q8) vraw[Q8 ? 4 * t + i : 0] = q8_load(vb + (int64_t) kv * a.v_nb1, dvg * 8);
                    else    vreg[Q8 ? 0 : 4 * t + i] = *(const u32x4 *)(vb + (int64_t) kv * a.v_nb1 + dvg * 16);
                }
            }
        }
    };
    auto store_tile = [&](int kv0) {
#pragma unroll
        for (int t = 0; t < NKC; t++) {
            const int c = tid + t * NT, row = c / CPR, col = c % CPR;
            if ((64 * CPR) % NT == 0 || c < 64 * CPR) *(u32x4 *)(Kl + row * (D * 2) + ((col ^ (row & (CPR - 1))) << 4)) = Q8 ? q8_cvt(kraw[Q8 ? t : 0]) : kreg[Q8 ? 0 : t];
        }
        if (VL == 1) {
#pragma unroll
            for (int t = 0; t < NVU; t++) {
                const int c = tid + t * NT, dv = c >> 3, kc = c & 7, kvs = kv0 + 8 * kc;
                u32x4 x = vreg[Q8 ? 0 : t];
                if (kv0 + 64 > n_kv) {                                 // (block-uniform: only the cache's last, partial tile)
                    const int left = n_kv - kvs;                       // elements of this chunk that exist (the rest of the cache row is not ours: may be anything)
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const uint32_t keep = 2 * w + 1 < left ? 0xffffffffu : 2 * w < left ? 0xffffu : 0u;
                        x[w] &= keep;
                    }
                }
                if ((D * 8) % NT == 0 || c < D * 8) { char * p = Vl + dv * FA_VSTR + kc * 16; *(u32x2 *) p = u32x2{x.x, x.y}; *(u32x2 *)(p + 8) = u32x2{x.z, x.w}; }
            }
        } else {
#pragma unroll
            for (int t = 0; t < NVU; t++) {
                const int u = tid + t * NT, kvg = u / CPR, dvg = u % CPR;
                u32x4 x[4];
#pragma unroll
                for (int i = 0; i < 4; i++) x[i] = Q8 ? q8_cvt(vraw[Q8 ? 4 * t + i : 0]) : vreg[Q8 ? 0 : 4 * t + i];
                if ((16 * CPR) % NT == 0 || u < 16 * CPR) {
#pragma unroll
                    for (int e = 0; e < 8; e++) {                      // 4 x 4 transposition in registers: V rows by position -> V^T rows by feature
                        const uint32_t sel = (e & 1) ? 0x07060302u : 0x05040100u;
                        const uint32_t lo = __builtin_amdgcn_perm(x[1][e >> 1], x[0][e >> 1], sel), up = __builtin_amdgcn_perm(x[3][e >> 1], x[2][e >> 1], sel);
                        *(u32x2 *)(Vl + (8 * dvg + e) * FA_VSTR + kvg * 8) = u32x2{lo, up};
                    }
                }
            }
        }
    };

    if (kv_lo < kv_hi) load_tile(kv_lo);
    for (int kv0 = kv_lo; kv0 < kv_hi; kv0 += 64) {
        // ---- the mask of this tile (tensor form): the lane's 32 positions in 8 groups of 4
        float mv[2][16];
        bool skip = false;
        if (MASK == 2) {
            int any = 0;
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int kvb = kv0 + 32 * j + 8 * g + 4 * hi;
                    uint16_t w[4];
                    if (a.m_al) {
                        u32x2 t = u32x2{0xfc00fc00u, 0xfc00fc00u};
                        if (kvb < kv_hi) t = *(const u32x2 *)(mrow + (int64_t) kvb * 2);
                        w[0] = (uint16_t) t.x; w[1] = (uint16_t)(t.x >> 16); w[2] = (uint16_t) t.y; w[3] = (uint16_t)(t.y >> 16);
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; e++) w[e] = kvb + e < kv_hi ? *(const uint16_t *)(mrow + (int64_t)(kvb + e) * 2) : (uint16_t) 0xfc00;
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const float f = kvb + e < kv_hi ? h2f(w[e]) : -INFINITY;
                        mv[j][4 * g + e] = f * mask_mul;                   // in units of the raw score (the scale is applied later)
                        any |= (f != -INFINITY) && rvalid;
                    }
                }
            skip = !__syncthreads_or(any);                             // (also: the previous tile's fragments have been read) nothing of this tile visible: no staging, no math
        } else {
            __syncthreads();                                           // the previous tile's fragments have been read
        }
        if (!skip) store_tile(kv0);
        if (kv0 + 64 < kv_hi) load_tile(kv0 + 64);                     // the next tile's loads fly under this tile's math
        if (skip) continue;
        __syncthreads();

        if (!wave_live) continue;                                      // (decode: the group's rows fit one wave; the others only help staging)
        // ---- S^T = K . Q^T.  The K fragments are read FA_PD MFMAs ahead of their use (the compiler's own order is read -> wait -> MFMA, one LDS
        //      latency per MFMA); sched_barrier pins the order
        f32x16 s[2];
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) s[j][r] = 0.0f;
        {
            auto kfrag = [&](int idx) {
                const int row = 32 * (idx / KS) + l31, ks = idx % KS;
                return *(const half8 *)(Kl + row * (D * 2) + (((2 * ks + hi) ^ (row & (CPR - 1))) << 4));
            };
            half8 kf[FA_PD];
#pragma unroll
            for (int p = 0; p < FA_PD; p++) kf[p] = kfrag(p);
#pragma unroll
            for (int idx = 0; idx < 2 * KS; idx++) {
                __builtin_amdgcn_sched_barrier(0);
                s[idx / KS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[idx % FA_PD], qf[idx % KS], s[idx / KS], 0, 0, 0);
                if (idx + FA_PD < 2 * KS) kf[idx % FA_PD] = kfrag(idx + FA_PD);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- online soft-max (base 2) of the row held by this lane pair.  The exponent reference m_run follows the running maximum lazily:
        //      it moves (and O, l are rescaled) only when a tile's maximum exceeds it by more than FA_TAU, so p <= 2^FA_TAU (fp16 holds it with the
        //      same relative precision) and the rescale of the 64 O registers leaves the steady state of the loop
        const bool full = MASK != 2 && kv0 + 64 <= kv_hi && (MASK == 0 || kv0 + 63 <= a.n_past + (qb * BQ + wave * 32) / a.R);      // nothing of this tile is masked for this wave
        float mx = -INFINITY;
        if (!full) {
#pragma unroll
            for (int j = 0; j < 2; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int kv = kv0 + 32 * j + fa_crow(r, hi);
                    float t = s[j][r];
                    if (MASK == 2) t += mv[j][r];                      // -inf beyond kv_hi
                    else if (MASK == 1) t = (kv <= lim && kv < kv_hi) ? t : -INFINITY;
                    else t = kv < kv_hi ? t : -INFINITY;
                    s[j][r] = t;
                }
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) mx = fmaxf(mx, s[j][r]);
        mx = fa_pair_max(mx) * a.sc2;                                  // sc2 > 0: the maximum of the scaled scores
        const bool grow = mx > m_run + FA_TAU;                         // (m_run = -inf: any finite maximum)
        if (__any(grow)) {
            const float m_new = grow ? mx : m_run;
            const float alpha = grow ? __builtin_amdgcn_exp2f(m_run - m_new) : 1.0f;      // m_run = -inf -> 0
            l_run *= alpha;
#pragma unroll
            for (int i = 0; i < NB; i++)
#pragma unroll
                for (int r = 0; r < 16; r++) o[i][r] *= alpha;
            m_run = m_new;
        }
        const float msafe = m_run == -INFINITY ? 0.0f : m_run;         // a row with nothing visible yet: every p = exp2(-inf) = 0
        float psum = 0.0f;
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) { const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], a.sc2, -msafe)); s[j][r] = p; psum += p; }
        l_run += psum;

        // ---- O^T += V^T . P^T: 4 NB MFMAs, V^T fragments read FA_PD ahead
        {
            half8 pb[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int e = 0; e < 8; e++) pb[q][e] = (_Float16) s[q >> 1][8 * (q & 1) + e];
            auto vfrag = [&](int idx) {                                // idx = 4-kv-group q (j, kk) major, dv block i minor
                const int q = idx / NB, i = idx % NB;
                const char * p = Vl + (32 * i + l31) * FA_VSTR + (32 * (q >> 1) + 16 * (q & 1) + 4 * hi) * 2;
                const half4 v0 = *(const half4 *) p, v1 = *(const half4 *)(p + 16);
                return half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            };
            half8 vf[FA_PD];
#pragma unroll
            for (int p = 0; p < FA_PD; p++) vf[p] = vfrag(p);
#pragma unroll
            for (int idx = 0; idx < 4 * NB; idx++) {
                __builtin_amdgcn_sched_barrier(0);
                o[idx % NB] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[idx % FA_PD], pb[idx / NB], o[idx % NB], 0, 0, 0);
                if (idx + FA_PD < 4 * NB) vf[idx % FA_PD] = vfrag(idx + FA_PD);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- epilogue: lane holds O^T[dv = 32 i + crow(r, hi), q = l31]: groups of 4 consecutive dv
    const float l_tot = fa_pair_sum(l_run);
    if (!rvalid) return;
    if (a.splits == 1) {
        const float inv = l_tot == 0.0f ? 0.0f : 1.0f / l_tot;
        char * dp = a.dst + (int64_t) n * a.d_nbn + (int64_t) h * a.d_nbh + (int64_t) b * a.d_nbb;
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *(f32x4 *)(dp + (32 * i + 8 * g + 4 * hi) * 4) = f32x4{o[i][4 * g] * inv, o[i][4 * g + 1] * inv, o[i][4 * g + 2] * inv, o[i][4 * g + 3] * inv};
    } else {
        float * pp = a.part + ((((int64_t) b * a.H + h) * a.N + n) * a.splits + split) * (D + 4);
#pragma unroll
        for (int i = 0; i < NB; i++)
#pragma unroll
            for (int g = 0; g < 4; g++)
                *(f32x4 *)(pp + 32 * i + 8 * g + 4 * hi) = f32x4{o[i][4 * g], o[i][4 * g + 1], o[i][4 * g + 2], o[i][4 * g + 3]};
        if (hi == 0) { pp[D] = m_run; pp[D + 1] = l_tot; }
    }
}

// merge the split-KV partials of one (b, h, n) (ggml_flash_attn_ext_reduce_partials, ops.cpp:8642-8710): 256 threads = D output elements x
// (256 / D) interleaved groups of splits, so that the loads of different splits are independent and coalesced
template <int D>
__global__ void __launch_bounds__(1024) k_fattn_merge(const fattn_args a) {
    constexpr int G = 1024 / D;
    __shared__ float red[1024];
    __shared__ float wgt[FA_MAX_SPLITS];
    __shared__ float Lsh;
    const int tid = threadIdx.x, dv = tid % D, g = tid / D, n = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const float * pp = a.part + (((int64_t) b * a.H + h) * a.N + n) * a.splits * (D + 4);
    float m = tid < a.splits ? pp[(int64_t) tid * (D + 4) + D] : -INFINITY, l = tid < a.splits ? pp[(int64_t) tid * (D + 4) + D + 1] : 0.0f;
    red[tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }      // splits <= 256
    const float M = red[0], Ms = M == -INFINITY ? 0.0f : M;
    __syncthreads();
    const float w = tid < a.splits ? __builtin_amdgcn_exp2f(m - Ms) : 0.0f;
    if (tid < FA_MAX_SPLITS) wgt[tid] = w;
    red[tid] = w * l;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    if (tid == 0) Lsh = red[0];
    __syncthreads();
    const float L = Lsh;
    __syncthreads();
    float acc = 0.0f;
#pragma unroll 4
    for (int s = g; s < a.splits; s += G) acc += wgt[s] * pp[(int64_t) s * (D + 4) + dv];
    red[tid] = acc;
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int k = 1; k < G; k++) acc += red[dv + k * D];
        const float inv = L == 0.0f ? 0.0f : 1.0f / L;
        float * dp = (float *)(a.dst + (int64_t) n * a.d_nbn + (int64_t) h * a.d_nbh + (int64_t) b * a.d_nbb);
        dp[dv] = acc * inv;
    }
}

static int g_flash_min = -1;             // -1: not read yet
int flash_prefill_min_cols() {
    if (g_flash_min < 0) {
        const char * off = getenv("CLLM_FLASH_PREFILL");
        const char * e = getenv("CLLM_MMA_MIN_COLS");          // the same threshold as the MFMA mat-muls it replaces (matmul_f.hip): <= 32 columns stay exact
        g_flash_min = off && atoi(off) == 0 ? 1 << 30 : e ? atoi(e) : 33;
    }
    return g_flash_min;
}
// tests: switch the runner's prefill attention between the flash kernel and the node sequence inside one process (<= 0: back to the environment)
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_attn_prefill_min_cols(int n) { g_flash_min = n > 0 ? n : -1; }
// split-KV partials exist only on the decode path (N * (H / Hkv) <= 32 query rows per KV head, so N <= 32): a prompt needs no scratch at all
// (it used to be sized for every N: 8.9 GB for a 2048-token -fa prefill that never touched it)
size_t fattn_wsize(int64_t N, int64_t H, int64_t B, int64_t D) { return N > 32 ? 0 : (size_t) B * H * N * FA_MAX_SPLITS * (D + 4) * 4; }

template <int D, int KVT, int VL, int MASK>
static int fattn_launch(hipStream_t st, const fattn_args & a, int B, bool decode) {
    const size_t lds = 64 * D * 2 + D * FA_VSTR;
    if (decode) {
        const dim3 grid((unsigned) a.splits, (unsigned)(a.H / a.R), (unsigned) B);
        hipLaunchKernelGGL((k_fattn<D, KVT, VL, MASK, 4>), grid, dim3(256), lds, st, a);     // four waves stage the tile (one load latency); the wave that owns the rows computes
        LAUNCH_CHECK();
        if (a.splits > 1) { hipLaunchKernelGGL((k_fattn_merge<D>), dim3((unsigned) a.N, (unsigned) a.H, (unsigned) B), dim3(1024), 0, st, a); LAUNCH_CHECK(); }
    } else {
        const dim3 grid((unsigned)((a.N + 127) / 128), (unsigned) a.H, (unsigned) B);
        hipLaunchKernelGGL((k_fattn<D, KVT, VL, MASK, 4>), grid, dim3(256), lds, st, a);
        LAUNCH_CHECK();
    }
    return CLLM_OK;
}

// q F32 [D, N, H, B]; k [D, n_kv, Hkv, B]; v: vl 0 [D, n_kv, Hkv, B], vl 1 [n_kv(ML), D, Hkv, B] (F16); mask F16 [n_kv, N, ..] or null;
// causal_past >= 0 (no mask tensor): position kv is visible to query n iff kv <= causal_past + n.
// dst element (dv, h, n, b) at dst.data + n * nbn + h * nbh + b * nbb.  CLLM_E_UNSUPPORTED: a shape / layout this kernel does not take.
int launch_fattn(hipStream_t st, const tview & q, const tview & k, int ktype, const tview & v, int vl, const tview * mask, int causal_past,
                 char * dst, int64_t nbn, int64_t nbh, int64_t nbb, float scale, void * wdata, size_t wsize) {
    const int64_t D = q.ne[0], N = q.ne[1], H = q.ne[2], B = q.ne[3], n_kv = k.ne[1], Hkv = k.ne[2];
    if (D != 64 && D != 128) return CLLM_E_UNSUPPORTED;
    if (k.ne[0] != D || (vl == 0 ? v.ne[0] != D || v.ne[1] != n_kv : v.ne[1] != D || v.ne[0] < n_kv) || v.ne[2] != Hkv || Hkv <= 0 || H % Hkv || k.ne[3] != B || v.ne[3] != B) return CLLM_E_UNSUPPORTED;
    if (ktype != CLLM_TYPE_F16 && ktype != CLLM_TYPE_Q8_0) return CLLM_E_UNSUPPORTED;
    if (vl == 1 && ktype != CLLM_TYPE_F16) return CLLM_E_UNSUPPORTED;
    if (N <= 0 || n_kv <= 0 || N > (1 << 24) || n_kv > (1 << 30) || H > 65535 || B > 65535) return CLLM_E_UNSUPPORTED;
    if (q.nb[0] != 4 || (((uintptr_t) q.data | (uintptr_t) q.nb[1] | (uintptr_t) q.nb[2] | (uintptr_t) q.nb[3]) & 15)) return CLLM_E_UNSUPPORTED;
    if ((((uintptr_t) dst | (uintptr_t) nbn | (uintptr_t) nbh | (uintptr_t) nbb) & 15)) return CLLM_E_UNSUPPORTED;
    if (ktype == CLLM_TYPE_F16) {
        if ((((uintptr_t) k.data | (uintptr_t) k.nb[1] | (uintptr_t) k.nb[2] | (uintptr_t) k.nb[3]) & 15)) return CLLM_E_UNSUPPORTED;
        if ((((uintptr_t) v.data | (uintptr_t) v.nb[1] | (uintptr_t) v.nb[2] | (uintptr_t) v.nb[3]) & 15)) return CLLM_E_UNSUPPORTED;
    } else {
        if ((((uintptr_t) k.data | (uintptr_t) k.nb[1] | (uintptr_t) k.nb[2] | (uintptr_t) k.nb[3] | (uintptr_t) v.data | (uintptr_t) v.nb[1] | (uintptr_t) v.nb[2] | (uintptr_t) v.nb[3]) & 1)) return CLLM_E_UNSUPPORTED;
    }
    fattn_args a;
    a.q = q.data; a.q_nb1 = q.nb[1]; a.q_nb2 = q.nb[2]; a.q_nb3 = q.nb[3];
    a.k = k.data; a.k_nb1 = k.nb[1]; a.k_nb2 = k.nb[2]; a.k_nb3 = k.nb[3];
    a.v = v.data; a.v_nb1 = v.nb[1]; a.v_nb2 = v.nb[2]; a.v_nb3 = v.nb[3];
    a.mask = nullptr; a.m_nb1 = a.m_nb2 = a.m_nb3 = 0; a.m_ne2 = a.m_ne3 = 1; a.m_al = 0;
    int mmode = 0;
    if (mask) {
        if (mask->nb[0] != 2 || mask->ne[0] < n_kv || mask->ne[1] < N || ((uintptr_t) mask->data & 1)) return CLLM_E_UNSUPPORTED;
        a.mask = mask->data; a.m_nb1 = mask->nb[1]; a.m_nb2 = mask->nb[2]; a.m_nb3 = mask->nb[3]; a.m_ne2 = (int) mask->ne[2]; a.m_ne3 = (int) mask->ne[3];
        a.m_al = n_kv % 4 == 0 && (((uintptr_t) mask->data | (uintptr_t) mask->nb[1] | (uintptr_t) mask->nb[2] | (uintptr_t) mask->nb[3]) & 7) == 0;
        mmode = 2;
    } else if (causal_past >= 0) mmode = 1;
    a.dst = dst; a.d_nbn = nbn; a.d_nbh = nbh; a.d_nbb = nbb;
    a.N = (int) N; a.H = (int) H; a.Hkv = (int) Hkv; a.n_kv = (int) n_kv; a.n_past = causal_past; a.sc2 = scale * FA_LOG2E;
    const int r = (int)(H / Hkv);
    const bool decode = N * r <= 32;
    a.R = decode ? r : 1;
    a.splits = 1; a.chunk = (int)((n_kv + 63) / 64 * 64); a.part = nullptr; a.n_kv_dev = nullptr;
    if (decode && n_kv > 64) {
        const int tiles = (int)((n_kv + 63) / 64);      // ~2 workgroups per CU, each walking `per` tiles with the next tile's loads in flight
        static const int div = getenv("CLLM_FA_DIV") ? atoi(getenv("CLLM_FA_DIV")) : 64;
        int per = tiles / div; if (per < 1) per = 1; if (per < (tiles + FA_MAX_SPLITS - 1) / FA_MAX_SPLITS) per = (tiles + FA_MAX_SPLITS - 1) / FA_MAX_SPLITS;
        a.chunk = per * 64; a.splits = (tiles + per - 1) / per;
        if (a.splits > 1 && !wdata) { a.splits = 1; a.chunk = (int)((n_kv + 63) / 64 * 64); }      // a caller without scratch (cllm_op_attn_prefill): one workgroup walks the whole cache
        if (a.splits > 1) {
            if (wsize < (size_t) B * H * N * a.splits * (D + 4) * 4 || ((uintptr_t) wdata & 15)) FAIL(CLLM_E_INVALID, "flash_attn_ext: wdata too small");
            a.part = (float *) wdata;
        }
    }
    if (vl == 1 && (v.nb[1] / 2) % 8) return CLLM_E_UNSUPPORTED;
#define FA_GO(D_, KVT_, VL_, M_) return fattn_launch<D_, KVT_, VL_, M_>(st, a, (int) B, decode)
#define FA_MASKS(D_, KVT_, VL_) do { if (mmode == 0) FA_GO(D_, KVT_, VL_, 0); if (mmode == 1) FA_GO(D_, KVT_, VL_, 1); FA_GO(D_, KVT_, VL_, 2); } while (0)
    if (D == 128) {
        if (ktype == CLLM_TYPE_Q8_0) FA_MASKS(128, CLLM_TYPE_Q8_0, 0);
        if (vl == 1) FA_MASKS(128, CLLM_TYPE_F16, 1);
        FA_MASKS(128, CLLM_TYPE_F16, 0);
    } else {
        if (ktype == CLLM_TYPE_Q8_0) FA_MASKS(64, CLLM_TYPE_Q8_0, 0);
        if (vl == 1) FA_MASKS(64, CLLM_TYPE_F16, 1);
        FA_MASKS(64, CLLM_TYPE_F16, 0);
    }
#undef FA_MASKS
#undef FA_GO
    return CLLM_E_UNSUPPORTED;
}

// ---- single-token attention above attn_long_threshold() cached positions inside a captured decode step (the runner's graph, the module's
//      launch list): RoPE of q and k + both cache writes in one small launch (the same arithmetic as k_attn_dec: bit-identical cache contents),
//      then the split-KV flash kernel with the number of positions read on the device, then the merge.  Tolerance tier (like attn_long.hip, which
//      it replaces unless CLLM_ATTN_LONG_FLASH=0): ~2x faster at 4K-16K positions.
template <int HD, int MODE>
__global__ void __launch_bounds__(HD) k_rope_kv_prep(const float * __restrict__ qkv, const int32_t * __restrict__ pos_dev, const float * __restrict__ rope_cs, int nh, int nkv,
                                                     uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache, int ML, float * __restrict__ qrot) {
    constexpr int half = HD / 2, off = MODE == 0 ? 1 : half;
    const int tid = threadIdx.x, blk = blockIdx.x, pos = pos_dev[0], KD = nkv * HD, QD = nh * HD;
    const bool is_q = blk < nh;
    const int g = blk - nh;
    const float * x = is_q ? qkv + blk * HD : qkv + QD + g * HD;
    if (tid < half) {
        const int ic = MODE == 0 ? 2 * tid : tid;
        const float x0 = x[ic], x1 = x[ic + off], c = rope_cs[2 * tid], sn = rope_cs[2 * tid + 1];
        const float y0 = rope_rot_a(x0, x1, c, sn), y1 = rope_rot_b(x0, x1, c, sn);
        if (is_q) { qrot[blk * HD + ic] = y0; qrot[blk * HD + ic + off] = y1; }
        else { k_cache[(int64_t) pos * KD + g * HD + ic] = f2h(y0); k_cache[(int64_t) pos * KD + g * HD + ic + off] = f2h(y1); }
    }
    if (!is_q) v_cache[((int64_t) g * HD + tid) * ML + pos] = f2h(qkv[QD + KD + g * HD + tid]);
}

int launch_attn_long_flash(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode,
                           uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * S, size_t s_bytes, float * att) {
    // opt-in (CLLM_ATTN_LONG_FLASH=1): this form keeps the flash kernel's own summation order (tolerance tier); the default beyond attn_long_threshold() is
    // attn_long.hip, which accumulates in the reference's order
    static const bool on = getenv("CLLM_ATTN_LONG_FLASH") && atoi(getenv("CLLM_ATTN_LONG_FLASH")) == 1;
    if (!on || !rope_cs || (hd != 64 && hd != 128) || nkv <= 0 || nh % nkv || (int64_t) nh / nkv > 32 || ML % 8 || ML > (1 << 30) || ((uintptr_t) S & 15)) return CLLM_E_UNSUPPORTED;
    const size_t q_bytes = ((size_t) nh * hd * 4 + 255) & ~(size_t) 255;
    if (s_bytes <= q_bytes) return CLLM_E_UNSUPPORTED;
    int max_splits = (int)((s_bytes - q_bytes) / ((size_t) nh * (hd + 4) * 4));
    if (max_splits > FA_MAX_SPLITS) max_splits = FA_MAX_SPLITS;
    if (max_splits < 4) return CLLM_E_UNSUPPORTED;
    float * qrot = S;
    if (hd == 128) { if (mode == 0) hipLaunchKernelGGL((k_rope_kv_prep<128, 0>), dim3(nh + nkv), dim3(128), 0, st, qkv, pos_dev, rope_cs, nh, nkv, k_cache, v_cache, (int) ML, qrot);
                     else           hipLaunchKernelGGL((k_rope_kv_prep<128, 2>), dim3(nh + nkv), dim3(128), 0, st, qkv, pos_dev, rope_cs, nh, nkv, k_cache, v_cache, (int) ML, qrot); }
    else           { if (mode == 0) hipLaunchKernelGGL((k_rope_kv_prep<64, 0>),  dim3(nh + nkv), dim3(64),  0, st, qkv, pos_dev, rope_cs, nh, nkv, k_cache, v_cache, (int) ML, qrot);
                     else           hipLaunchKernelGGL((k_rope_kv_prep<64, 2>),  dim3(nh + nkv), dim3(64),  0, st, qkv, pos_dev, rope_cs, nh, nkv, k_cache, v_cache, (int) ML, qrot); }
    LAUNCH_CHECK();
    const int KD = nkv * hd;
    fattn_args a;
    a.q = (const char *) qrot; a.q_nb1 = (int64_t) nh * hd * 4; a.q_nb2 = (int64_t) hd * 4; a.q_nb3 = a.q_nb1;
    a.k = (const char *) k_cache; a.k_nb1 = (int64_t) KD * 2; a.k_nb2 = (int64_t) hd * 2; a.k_nb3 = a.k_nb1 * ML;
    a.v = (const char *) v_cache; a.v_nb1 = ML * 2; a.v_nb2 = ML * hd * 2; a.v_nb3 = a.v_nb2 * nkv;
    a.mask = nullptr; a.m_nb1 = a.m_nb2 = a.m_nb3 = 0; a.m_ne2 = a.m_ne3 = 1; a.m_al = 0;
    a.dst = (char *) att; a.d_nbn = (int64_t) nh * hd * 4; a.d_nbh = (int64_t) hd * 4; a.d_nbb = a.d_nbn;
    a.part = (float *)((char *) S + q_bytes);
    a.N = 1; a.H = nh; a.Hkv = nkv; a.R = nh / nkv; a.n_kv = (int) ML; a.n_past = -1; a.sc2 = (1.0f / sqrtf((float) hd)) * FA_LOG2E;
    a.n_kv_dev = pos_dev;
    const int tiles = (int)((ML + 63) / 64);
    int per = (tiles + max_splits - 1) / max_splits; if (per < tiles / 32) per = tiles / 32; if (per < 1) per = 1;
    a.chunk = per * 64; a.splits = (tiles + per - 1) / per;
    const size_t lds = 64 * (size_t) hd * 2 + (size_t) hd * FA_VSTR;
    const dim3 grid((unsigned) a.splits, (unsigned) nkv, 1);
    if (hd == 128) hipLaunchKernelGGL((k_fattn<128, CLLM_TYPE_F16, 1, 0, 4>), grid, dim3(256), lds, st, a);
    else           hipLaunchKernelGGL((k_fattn<64,  CLLM_TYPE_F16, 1, 0, 4>), grid, dim3(256), lds, st, a);
    LAUNCH_CHECK();
    if (hd == 128) hipLaunchKernelGGL((k_fattn_merge<128>), dim3(1, (unsigned) nh, 1), dim3(1024), 0, st, a);
    else           hipLaunchKernelGGL((k_fattn_merge<64>),  dim3(1, (unsigned) nh, 1), dim3(1024), 0, st, a);
    LAUNCH_CHECK();
    return CLLM_OK;
}
