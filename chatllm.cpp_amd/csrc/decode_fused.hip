// decode_fused.hip -- fused single-token (decode) kernels used by the decoder runner (decoder.hip).
//
// Same arithmetic, operation order and rounding points as the individual graph nodes they replace:
//   k_attn_decode     [ROPE(q), ROPE(k) + SET_ROWS(K -> f16 cache row pos) + CPY(V^T -> f16 cache column pos)] +
//                     MUL_MAT(K,Q) + SCALE + DIAG_MASK_INF + SOFT_MAX + MUL_MAT(V,P) [+ PERMUTE/CONT] for one query token
//   k_argmax_*        greedy sampler (std::max_element, first maximum) feeding the next step on the device
// (RMS_NORM + MUL, SiLU * up and the activation quantization of MUL_MAT's src1 are prologues of the GEMV, mmvq.hip.)
// The only per-token inputs (token id, position) are read from device memory, so a whole decode step is a static
// launch sequence that the runner captures once in a hipGraph.
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"
#include "dequant.h"

#include <math.h>

// GGML_F32x8_REDUCE (simd-mappings.h:545-560) over the 32 accumulators of ggml_vec_dot_f16, held two per lane by 16 lanes (accumulator
// a = 8 j + l in lane a / 2): x0 += x2, x1 += x3 (lanes c ^ 8); x0 += x1 (c ^ 4); lo + hi halves (c ^ 2); hadd, hadd (in-lane, c ^ 1)
__device__ __forceinline__ float lane_xor4_ff(float v) { return __int_as_float(lane_xor4_i(__float_as_int(v))); }
__device__ __forceinline__ float vd32_reduce(float a0, float a1) {
    a0 = a0 + dpp_f<DPP_ROW_ROR8>(a0); a1 = a1 + dpp_f<DPP_ROW_ROR8>(a1);
    a0 = a0 + lane_xor4_ff(a0);        a1 = a1 + lane_xor4_ff(a1);
    a0 = a0 + dpp_f<DPP_QUAD_XOR2>(a0); a1 = a1 + dpp_f<DPP_QUAD_XOR2>(a1);
    const float u = a0 + a1;
    return u + dpp_f<DPP_QUAD_XOR1>(u);
}

// ---------------------------------------------------------------------------------------------------------------
// Attention for one query token, one workgroup per query head.
//   scores[i] = sum_d f16(K[i][d]) * f16r(q[d])          ggml_vec_dot_f16 semantics: src1 rounded to fp16, fp32 accumulate
//   p = soft_max(scores * scale)  (every cached position is visible to the newest token: the causal mask is empty)
//       ggml_vec_soft_max_f32: groups of 8 through ggml_v_expf, f32 tree sum per group, total in double, expf tail
//   ctx[d] = sum_i f16(V[d][i]) * f16r(p[i])
// ---------------------------------------------------------------------------------------------------------------
// ROPE = true: q and k arrive UN-rotated in qkv; every workgroup rotates its own q head and its group's k head, rounds
// the new k / v to fp16 exactly as the cache write would, uses them straight from LDS for position `pos`, and the first
// head of each GQA group stores them into the caches (so RoPE + SET_ROWS + CPY cost no launch of their own).
static float * g_attn_dbg = nullptr;
static unsigned long long * g_attn_ts = nullptr;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_attn_ts(unsigned long long * dev_buf) { g_attn_ts = dev_buf; }   // tools only
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_attn_probs(float * dev_buf) { g_attn_dbg = dev_buf; }   // tools only

// Latency structure (one launch is ~4.5 us of fixed cost, the rest is a chain of dependent memory round trips): the
// loads of this head's q / k / v projections are issued first, then the first batch of K-cache rows and V-cache rows
// (they depend only on the position), and only then does the workgroup wait for the projections and do the RoPE /
// rounding work -- the cache latencies overlap it.  1024 threads: 64 lane groups keep 256 cache rows in flight.
template <bool ROPE>
__global__ void __launch_bounds__(1024) k_attn_decode(const float * __restrict__ qkv, const int32_t * __restrict__ pos_dev, int nh, int nkv, int hd,
                                                      float scale, uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache,
                                                      int64_t ML, float * __restrict__ att, float * __restrict__ dbg, int mode, float theta_scale) {
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [hd] q (fp16-rounded) | [hd] new k | [hd] new v | [n_kv] scores
    __shared__ double red_d[1];
    __shared__ float  red_f[16];
    const int h = blockIdx.x, r2 = nh / nkv, g = h / r2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x, nw = nthr >> 6;
    const int KD = nkv * hd, QD = nh * hd;
    const int half = hd / 2;
    float * qs = sm; float * knew = sm + hd; float * vnew = sm + 2 * hd; float * sc = sm + 3 * hd;

    // ---- (1) this head's projections: one (q or k) pair or one v element per thread, loads issued before anything waits ----
    const int off = mode == 0 ? 1 : half;
    const float * qh = qkv + h * hd; const float * kh = qkv + QD + g * hd; const float * vh = qkv + QD + KD + g * hd;
    float px0 = 0.0f, px1 = 0.0f, pv = 0.0f;
    const bool pair_fast = ROPE && 2 * half + hd <= nthr;             // every pair / v element has its own thread
    if (pair_fast) {
        if (tid < 2 * half) {
            const int which = tid / half, i = tid % half, ic = mode == 0 ? 2*i : i;
            const float * x = which == 0 ? qh : kh;
            px0 = x[ic]; px1 = x[ic + off];
        } else if (tid < 2 * half + hd) pv = vh[tid - 2 * half];
    }
    const int pos = pos_dev[0];
    const int n_kv = pos + 1;

    // ---- (2) lane grouping: 16 lanes per cache row, accumulators of ggml_vec_dot_f16 two per lane (see k_attn_dec below) ----
    const int c16 = lane & 15, sub = lane >> 4, nrg = nw * 4;
    const int hnp = hd & ~31;

    // ---- (3) RoPE, fp16 rounding, cache write ----
    if (ROPE) {
        if (pair_fast) {
            if (tid < 2 * half) {
                const int which = tid / half, i = tid % half, ic = mode == 0 ? 2*i : i;
                float theta = (float) pos;
                for (int k = 0; k < i; k++) theta *= theta_scale;
                float c, s_;
                rope_cos_sin(theta, &c, &s_);
                c = c * 1.0f; s_ = s_ * 1.0f;
                const float y0 = rope_rot_a(px0, px1, c, s_), y1 = rope_rot_b(px0, px1, c, s_);
                float * o = which == 0 ? qs : knew;
                o[ic] = h2f(f2h(y0)); o[ic + off] = h2f(f2h(y1));         // q: src1 of K.Q is rounded to fp16; k: the cache is fp16
            } else if (tid < 2 * half + hd) vnew[tid - 2 * half] = h2f(f2h(pv));
        } else {
            float * cs = sc;                                            // scores are not live yet
            for (int i = tid; i < half; i += nthr) {
                float theta = (float) pos;
                for (int k = 0; k < i; k++) theta *= theta_scale;
                float c, s_;
                rope_cos_sin(theta, &c, &s_);
                cs[2*i] = c * 1.0f; cs[2*i + 1] = s_ * 1.0f;
            }
            __syncthreads();
            for (int t = tid; t < 2 * half; t += nthr) {
                const int which = t / half, i = t % half, ic = mode == 0 ? 2*i : i;
                const float c = cs[2*i], s_ = cs[2*i + 1];
                const float * x = which == 0 ? qh : kh;
                const float x0 = x[ic], x1 = x[ic + off];
                const float y0 = rope_rot_a(x0, x1, c, s_), y1 = rope_rot_b(x0, x1, c, s_);
                float * o = which == 0 ? qs : knew;
                o[ic] = h2f(f2h(y0)); o[ic + off] = h2f(f2h(y1));
            }
            for (int d = tid; d < hd; d += nthr) vnew[d] = h2f(f2h(vh[d]));
        }
        __syncthreads();
        if (h % r2 == 0) {
            for (int d = tid; d < hd; d += nthr) {
                k_cache[(int64_t) pos * KD + g * hd + d] = f2h(knew[d]);
                v_cache[((int64_t) g * hd + d) * ML + pos] = f2h(vnew[d]);
            }
        }
    } else {
        for (int d = tid; d < hd; d += nthr) qs[d] = h2f(f2h(qkv[h * hd + d]));
        __syncthreads();
    }

    // ---- (4) scores[i] = K[i] . q * scale, in ggml_vec_dot_f16's order ----
    for (int i0 = wave * 4 + sub; i0 < n_kv; i0 += nrg) {
        const uint16_t * kr = k_cache + (int64_t) i0 * KD + g * hd;
        const bool fr = ROPE && i0 == pos;
        float a0 = 0.0f, a1 = 0.0f;
        for (int e = 2 * c16; e < hnp; e += 32) {
            const float k0 = fr ? knew[e] : h2f(kr[e]), k1 = fr ? knew[e + 1] : h2f(kr[e + 1]);
            a0 = __builtin_fmaf(k0, qs[e], a0); a1 = __builtin_fmaf(k1, qs[e + 1], a1);
        }
        float v = vd32_reduce(a0, a1);
        if (c16 == 0) {
            if (hnp < hd) { double s = (double) v; for (int e = hnp; e < hd; e++) s += (double)((fr ? knew[e] : h2f(kr[e])) * qs[e]); v = (float) s; }
            sc[i0] = v * scale;                                           // the SCALE node
        }
    }
    __syncthreads();

    // ---- (5) soft_max over sc[0..n_kv): same partition as k_soft_max (one wave, lane = groups of 8) ----
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += nthr) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    for (int w = 1; w < nw; w++) mx = fmaxf(mx, red_f[w]);
    const int nv = n_kv & ~7;                  // (all threads compute the exponentials; the sum keeps the reference's order: see k_attn_dec)
    for (int i = tid; i < n_kv; i += nthr) sc[i] = i < nv ? ggml_expf_poly(sc[i] - mx) : libm_expf(sc[i] - mx);
    __syncthreads();
    if (wave == 0) {
        double sum = 0.0;
        for (int gi = lane * 8; gi < nv; gi += 64 * 8) {
            const f32x4 lo = *(const f32x4 *)(sc + gi), up = *(const f32x4 *)(sc + gi + 4);
            const float a0 = lo.x + up.x, a1 = lo.y + up.y, a2 = lo.z + up.z, a3 = lo.w + up.w;
            sum += (double)((a0 + a2) + (a1 + a3));
        }
        if (lane == 0) for (int i = nv; i < n_kv; i++) sum += (double) sc[i];
        sum = wave_sum_d(sum);
        double rinv = 1.0 / sum;                                      // (ORDER: soft_total_order_safe, common.h)
        if (__builtin_expect(!soft_total_order_safe(rinv, n_kv >> 3), 0)) rinv = 1.0 / soft_sum_serial(sc, nv, n_kv);
        if (lane == 0) red_d[0] = rinv;
    }
    __syncthreads();
    const float inv = (float) red_d[0];
    for (int i = tid; i < n_kv; i += nthr) sc[i] = h2f(f2h(sc[i] * inv));   // probability, then its fp16 rounding for V.P
    __syncthreads();
    if (dbg) { for (int i = tid; i < n_kv; i += nthr) dbg[(int64_t) h * ML + i] = sc[i]; if (tid == 0) { dbg[(int64_t) nh * ML + 2*h] = mx; dbg[(int64_t) nh * ML + 2*h + 1] = inv; } }

    // ---- (6) ctx = V . P in the same order: chunks of 32 cached positions, then the n_kv mod 32 leftovers in double ----
    float * tailp = sm + 3 * hd + (ML > hd ? ML : hd) + (wave * 4 + sub) * 32;
    const int np = n_kv & ~31, ntail = n_kv - np;
    for (int d0 = wave * 4 + sub; d0 < hd; d0 += nrg) {
        const uint16_t * vr = v_cache + ((int64_t) g * hd + d0) * ML;
        const float vfresh = ROPE ? vnew[d0] : 0.0f;
        float a0 = 0.0f, a1 = 0.0f;
        for (int e = 2 * c16; e < np; e += 32) {
            const float v0 = (ROPE && e == pos) ? vfresh : h2f(vr[e]), v1 = (ROPE && e + 1 == pos) ? vfresh : h2f(vr[e + 1]);
            a0 = __builtin_fmaf(v0, sc[e], a0); a1 = __builtin_fmaf(v1, sc[e + 1], a1);
        }
        const float res = vd32_reduce(a0, a1);
        for (int t = 0; t < 2; t++) {
            const int e = np + 2 * c16 + t;
            if (e < n_kv) tailp[2 * c16 + t] = ((ROPE && e == pos) ? vfresh : h2f(vr[e])) * sc[e];
        }
        wave_lds_fence();
        if (c16 == 0) {
            double s = (double) res;
            for (int t = 0; t < ntail; t++) s += (double) tailp[t];
            att[h * hd + d0] = (float) s;
        }
        wave_lds_fence();
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same attention for the common head sizes (64 / 128), as SMALL code: a launch starts with a cold instruction cache
// (~1 us per KB of straight-line code on the critical path, tools/gemv_phase_probe.py), and the general kernel above is
// 25 KB.  Compile-time head size and RoPE pairing, no integer divisions (the GQA group comes from blockIdx.y), and the
// cos/sin of the position come from a table built once per token (k_rope_table) instead of once per head and layer.
// Arithmetic and summation order are those of the general kernel (= of the unfused graph nodes).
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_rope_table(const int32_t * __restrict__ pos_dev, int half, float theta_scale, float * __restrict__ cs) {
    for (int i = threadIdx.x; i < half; i += 64) {
        float theta = (float) pos_dev[0];
        for (int k = 0; k < i; k++) theta *= theta_scale;        // iterated fp32 multiplication, ops.cpp:5639-5650
        float c, s_;
        rope_cos_sin(theta, &c, &s_);
        cs[2*i] = c * 1.0f; cs[2*i + 1] = s_ * 1.0f;
    }
}
int launch_rope_table(hipStream_t st, const int32_t * pos_dev, int hd, float freq_base, float * cs) {
    hipLaunchKernelGGL(k_rope_table, dim3(1), dim3(64), 0, st, pos_dev, hd / 2, powf(freq_base, -2.0f / hd), cs);
    LAUNCH_CHECK();
    return CLLM_OK;
}

__device__ __forceinline__ int uniform_load_i32(const int32_t * p) {
    int v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

template <int HD, int MODE, int PARTS>      // MODE 0: adjacent pairs (GGML_ROPE_TYPE_NORMAL), 2: NEOX halves; PARTS: workgroups per head (each redoes RoPE, scores and
                                            // soft_max -- the same bits -- and takes 1 / PARTS of the V.P rows: one row per 16-lane group instead of two)
__global__ void __launch_bounds__(1024) k_attn_dec(const float * __restrict__ qkv, const int32_t * __restrict__ pos_dev, const float * __restrict__ rope_cs,
                                                   int nh, int nkv, float scale, uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache,
                                                   int ML, float * __restrict__ att, unsigned long long * ts) {
#define TS(k) do { if (ts && threadIdx.x == 0) ts[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (k)] = wall_clock64(); } while (0)
    extern __shared__ __attribute__((aligned(16))) float sm[];       // [HD] q (fp16-rounded) | [HD] new k | [HD] new v | [n_kv] scores
    __shared__ float  red_f[16];
    __shared__ uint64_t etab_s[32];                                   // glibc's exp2f table for the soft_max's n mod 8 leftovers (see (5))
    constexpr int half = HD / 2, off = MODE == 0 ? 1 : half, U = 4;
    const int r2 = gridDim.x, g = blockIdx.y, h = g * r2 + blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int KD = nkv * HD, QD = nh * HD;
    float * qs = sm; float * knew = sm + HD; float * vnew = sm + 2 * HD; float * sc = sm + 3 * HD;

    // ---- (1) this head's projections + the cos/sin of its pairs: loads issued before anything waits ----
    float px0 = 0.0f, px1 = 0.0f, pc = 0.0f, ps = 0.0f;
    const bool is_pair = tid < 2 * half, is_v = !is_pair && tid < 2 * half + HD;
    const int which = tid >= half ? 1 : 0, pi = tid - which * half, ic = MODE == 0 ? 2 * pi : pi;
    if (is_pair) {
        const float * x = which == 0 ? qkv + h * HD : qkv + QD + g * HD;
        px0 = x[ic]; px1 = x[ic + off];
        pc = rope_cs[2 * pi]; ps = rope_cs[2 * pi + 1];
    } else if (is_v) px0 = qkv[QD + KD + g * HD + (tid - 2 * half)];
    const int pos = uniform_load_i32(pos_dev);
    const int n_kv = pos + 1;
    uint64_t et = 0;
    if (tid >= 992) et = gm_exp2f_T_dev[tid - 992];                   // half a wave fetches the table with the first loads; it is in LDS behind the RoPE barrier

    // ---- (2) first batch of cache rows.  Accumulation ORDER of ggml_vec_dot_f16 (vec.cpp:264-, AVX2 + F16C: simd-mappings.h:528-620), so
    //          that scores and context equal the reference's bit for bit: 32 fp32 accumulators, accumulator a takes elements a, a + 32, ...
    //          in order (one fma each), GGML_F32x8_REDUCE's tree, leftovers (n mod 32) one by one in double.  16 lanes per row, lane c
    //          carries accumulators 2c and 2c + 1: one dword (two fp16) per 32-element chunk. ----
    constexpr int NCH = HD / 32, VU = HD / 64 / PARTS, VPF = 8;
    const int vrow0 = PARTS > 1 ? (int) blockIdx.z * VU * 64 : 0;       // first V^T row of this part
    const int c16 = lane & 15, sub = lane >> 4;
    const int ib0 = wave * 4 + sub;                                   // K rows ib0 + 64 u; V^T rows ib0 + 64 u
    uint32_t kr0[U][NCH];
#pragma unroll
    for (int u = 0; u < U; u++) {
        const int i0 = ib0 + u * 64;
#pragma unroll
        for (int i = 0; i < NCH; i++)        // unconditional (clamped) loads: branches around loads cost the compiler its vmcnt bookkeeping (it then waits with vmcnt(0))
            kr0[u][i] = *(const uint32_t *)(k_cache + (int64_t)(i0 < pos ? i0 : 0) * KD + g * HD + 32 * i + 2 * c16);      // row `pos` is the new k (LDS)
    }
    const int np = n_kv & ~31, nch = np >> 5;
    uint32_t vc0[VU][VPF];
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const uint16_t * vr = v_cache + ((int64_t) g * HD + vrow0 + ib0 + u * 64) * ML;
#pragma unroll
        for (int i = 0; i < VPF; i++) vc0[u][i] = *(const uint32_t *)(vr + 32 * (i < nch ? i : 0) + 2 * c16);
    }

    TS(0);
    // ---- (3) RoPE, fp16 rounding, cache write ----
    if (is_pair) {
        const float y0 = rope_rot_a(px0, px1, pc, ps), y1 = rope_rot_b(px0, px1, pc, ps);
        float * o = which == 0 ? qs : knew;
        o[ic] = h2f(f2h(y0)); o[ic + off] = h2f(f2h(y1));             // q: src1 of K.Q is rounded to fp16; k: the cache is fp16
    } else if (is_v) vnew[tid - 2 * half] = h2f(f2h(px0));
    if (tid >= 992) etab_s[tid - 992] = et;
    lds_barrier();
    if (blockIdx.x == 0 && (PARTS == 1 || blockIdx.z == 0) && tid < HD) {
        k_cache[(int64_t) pos * KD + g * HD + tid] = f2h(knew[tid]);
        v_cache[((int64_t) g * HD + tid) * ML + pos] = f2h(vnew[tid]);
    }

    TS(1);
    // ---- (4) scores[i] = K[i] . q * scale.  Small code on purpose (a launch starts with a cold instruction cache): the cached rows i < pos in one branch-free loop body,
    //          the batch in registers on entry (the first one was requested at kernel entry), the next one requested behind the math; the new row (pos: in LDS, the cache
    //          write may not have landed) by one 16-lane group afterwards, in the same order ----
    {
        float qv[NCH][2];
#pragma unroll
        for (int i = 0; i < NCH; i++) { qv[i][0] = qs[32 * i + 2 * c16]; qv[i][1] = qs[32 * i + 2 * c16 + 1]; }
        if (wave == 15 && sub == 3) {                                     // the new row first: its operands are in LDS, the cached rows may still be in flight
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int i = 0; i < NCH; i++) { a0 = __builtin_fmaf(knew[32 * i + 2 * c16], qv[i][0], a0); a1 = __builtin_fmaf(knew[32 * i + 2 * c16 + 1], qv[i][1], a1); }
            const float v = vd32_reduce(a0, a1);
            if (c16 == 0) sc[pos] = v * scale;
        }
#pragma clang loop unroll(disable)
        for (int ib = ib0; ib < pos; ib += U * 64) {
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i0 = ib + u * 64;
                float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
                for (int i = 0; i < NCH; i++) {
                    a0 = __builtin_fmaf(h2f((uint16_t)(kr0[u][i] & 0xffff)), qv[i][0], a0); a1 = __builtin_fmaf(h2f((uint16_t)(kr0[u][i] >> 16)), qv[i][1], a1);
                }
                const float v = vd32_reduce(a0, a1);
                if (c16 == 0 && i0 < pos) sc[i0] = v * scale;             // the SCALE node
            }
            if (ib + U * 64 < pos) {
#pragma unroll
                for (int u = 0; u < U; u++) {
                    const int i0 = ib + (U + u) * 64;
#pragma unroll
                    for (int i = 0; i < NCH; i++) kr0[u][i] = *(const uint32_t *)(k_cache + (int64_t)(i0 < pos ? i0 : 0) * KD + g * HD + 32 * i + 2 * c16);
                }
            }
        }
    }
    lds_barrier();
    TS(2);

    // ---- the next VPF chunks of the V^T rows and the leftover elements are requested here: the soft_max hides their latency (contexts up to
    //      16 chunks = 512 positions never wait in (6); small code on purpose: a launch starts with a cold instruction cache) ----
    uint32_t vc1[VU][VPF]; uint16_t vtl[VU][2];
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const uint16_t * vr = v_cache + ((int64_t) g * HD + vrow0 + ib0 + u * 64) * ML;
#pragma unroll
        for (int i = 0; i < VPF; i++) vc1[u][i] = *(const uint32_t *)(vr + 32 * (VPF + i < nch ? VPF + i : 0) + 2 * c16);
#pragma unroll
        for (int t = 0; t < 2; t++) { const int e = np + 2 * c16 + t; vtl[u][t] = vr[e < n_kv ? e : 0]; }
    }
    // ---- (5) soft_max over sc[0..n_kv): same partition as k_soft_max (one wave, lane = groups of 8) ----
    float mx = -INFINITY;
    for (int i = tid; i < n_kv; i += 1024) mx = fmaxf(mx, sc[i]);
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    lds_barrier();
    mx = red_f[0];
#pragma unroll
    for (int w = 1; w < 16; w++) mx = fmaxf(mx, red_f[w]);
    // every exponential is independent: all threads compute them (the groups of 8 through the AVX2 polynomial, the n_kv mod 8 leftovers through
    // expf, as ggml_vec_soft_max_f32 does); the total is one wave's (lane = groups of 8 -> double; leftovers last)
    const int nv = n_kv & ~7;
    for (int i = tid; i < n_kv; i += 1024) {
        const float xv = sc[i] - mx;
        float ev;
        if (i < nv) ev = ggml_expf_poly(xv);
        else { const uint64_t ki = gm_expf_ki(xv); ev = gm_expf_fin(xv, ki, etab_s[ki & 31]); }      // glibc's expf, inline, its table entry from LDS: a call + a constant-memory
        sc[i] = ev;                                                                                  // load behind the maximum cost ~0.7 us in 7 steps of 8
    }
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
        // (ORDER: a tree, the reference adds serially -- proved not to matter per row, else redone serially: soft_total_order_safe, common.h)
        double rinv = 1.0 / sum;
        if (__builtin_expect(!soft_total_order_safe(rinv, n_kv >> 3), 0)) rinv = 1.0 / soft_sum_serial(sc, nv, n_kv);
        if (lane == 0) red_f[0] = (float) rinv;                       // (every thread read the maxima before the barrier above)
    }
    lds_barrier();
    const float inv = red_f[0];
    for (int i = tid; i < n_kv; i += 1024) sc[i] = h2f(f2h(sc[i] * inv));   // probability, then its fp16 rounding for V.P
    lds_barrier();
    TS(3);

    // ---- (6) ctx = V . P: chunks of 32 cached positions in order, then the n_kv mod 32 leftovers in double (products staged in LDS).  Small code: one body per chunk,
    //          the first 8 chunks from vc0 (requested at kernel entry), the next 8 from vc1 (requested behind the scores), a plain loop beyond ----
    float * tailp = sc + ML + (wave * 4 + sub) * 32;
    const int ntail = n_kv - np;
#pragma unroll
    for (int u = 0; u < VU; u++) {
        const int d0 = vrow0 + ib0 + u * 64;
        const uint16_t * vr = v_cache + ((int64_t) g * HD + d0) * ML;
        const float vfresh = vnew[d0];
        float a0 = 0.0f, a1 = 0.0f;
        auto chunk = [&](uint32_t w, int ci) {
            const int e = 32 * ci + 2 * c16;
            typedef float f32x2_t __attribute__((ext_vector_type(2)));
            const f32x2_t pr = *(const f32x2_t *)(sc + e);
            const float v0 = e == pos ? vfresh : h2f((uint16_t)(w & 0xffff)), v1 = e + 1 == pos ? vfresh : h2f((uint16_t)(w >> 16));
            a0 = __builtin_fmaf(v0, pr.x, a0); a1 = __builtin_fmaf(v1, pr.y, a1);
        };
#pragma unroll
        for (int i = 0; i < VPF; i++) if (i < nch) chunk(vc0[u][i], i);                    // (uniform branches: nch comes from a scalar load)
        if (nch > VPF) {
#pragma unroll
            for (int i = 0; i < VPF; i++) if (VPF + i < nch) chunk(vc1[u][i], VPF + i);
#pragma clang loop unroll(disable)
            for (int c8 = 2 * VPF; c8 < nch; c8 += VPF) {                                  // beyond 512 cached positions (attn_long.hip's regime by default): loads, then the math
                uint32_t w[VPF];
#pragma unroll
                for (int i = 0; i < VPF; i++) w[i] = *(const uint32_t *)(vr + 32 * (c8 + i < nch ? c8 + i : 0) + 2 * c16);
#pragma unroll
                for (int i = 0; i < VPF; i++) if (c8 + i < nch) chunk(w[i], c8 + i);
            }
        }
        const float res = vd32_reduce(a0, a1);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int e = np + 2 * c16 + t;
            if (e < n_kv) tailp[2 * c16 + t] = (e == pos ? vfresh : h2f(vtl[u][t])) * sc[e];
        }
        wave_lds_fence();
        if (c16 == 0) {
            double s = (double) res;
            for (int t = 0; t < ntail; t++) s += (double) tailp[t];
            att[h * HD + d0] = (float) s;
        }
        wave_lds_fence();
    }
    TS(4);
#undef TS
}

// RoPE + KV-cache write + attention with the position's cos/sin table (launch_rope_table); CLLM_E_UNSUPPORTED -> use the general kernel
int launch_attn_dec_table(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode,
                          uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * att) {
    if ((hd != 64 && hd != 128) || nh % nkv || ML % 8 || ML > (1 << 30) || g_attn_dbg) return CLLM_E_UNSUPPORTED;
    const size_t lds = (size_t)(3 * hd + ML) * 4 + 64 * 32 * 4;      // q | new k | new v | scores | leftover products of 64 lane groups
    if (lds > 150 * 1024) return CLLM_E_UNSUPPORTED;
    const float scale = 1.0f / sqrtf((float) hd);
    const dim3 grid(nh / nkv, nkv, hd == 128 ? 2 : 1);
#define GO(HD_, MODE_) do { \
        static uint64_t attr = 0; \
        if (lds > 48 * 1024 && dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_dec<HD_, MODE_, (HD_ == 128 ? 2 : 1)>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_attn_dec<HD_, MODE_, (HD_ == 128 ? 2 : 1)>), grid, dim3(1024), lds, st, qkv, pos_dev, rope_cs, nh, nkv, scale, k_cache, v_cache, (int) ML, att, g_attn_ts); } while (0)
    if (hd == 128) { if (mode == 0) GO(128, 0); else GO(128, 2); }      // any other mode pairs NEOX-style, as in the general kernel
    else           { if (mode == 0) GO(64, 0);  else GO(64, 2); }
#undef GO
    LAUNCH_CHECK();
    return CLLM_OK;
}

static int attn_launch(hipStream_t st, bool rope, const float * qkv, const int32_t * pos_dev, int nh, int nkv, int hd, uint16_t * k_cache, uint16_t * v_cache,
                       int64_t ML, float * att, int mode, float freq_base) {
    if (hd % 8 || (ML % 8) || nh % nkv) FAIL(CLLM_E_UNSUPPORTED, "attn_decode: head_dim and max_len must be multiples of 8");
    const size_t lds = (size_t)(3 * hd + (ML > hd ? ML : hd)) * 4 + 64 * 32 * 4;      // q | new k | new v | scores (reused for the cos/sin table) | leftover products
    if (lds > 150 * 1024) FAIL(CLLM_E_UNSUPPORTED, "attn_decode: max_len %lld does not fit LDS", (long long) ML);
    static uint64_t attr0 = 0, attr1 = 0;
    if (lds > 48 * 1024) {
        if (!rope && dev_flag_unset(attr0)) { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_decode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr0); }
        if (rope && dev_flag_unset(attr1))  { HIP_TRY(hipFuncSetAttribute((const void *) k_attn_decode<true>,  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr1); }
    }
    const float scale = 1.0f / sqrtf((float) hd), theta_scale = powf(freq_base, -2.0f / hd);
    if (rope) hipLaunchKernelGGL(k_attn_decode<true>,  dim3(nh), dim3(1024), lds, st, qkv, pos_dev, nh, nkv, hd, scale, k_cache, v_cache, ML, att, g_attn_dbg, mode, theta_scale);
    else      hipLaunchKernelGGL(k_attn_decode<false>, dim3(nh), dim3(1024), lds, st, qkv, pos_dev, nh, nkv, hd, scale, k_cache, v_cache, ML, att, g_attn_dbg, mode, theta_scale);
    LAUNCH_CHECK();
    return CLLM_OK;
}
int launch_attn_decode(hipStream_t st, const float * qkv, const int32_t * pos_dev, int nh, int nkv, int hd, const uint16_t * k_cache,
                       const uint16_t * v_cache, int64_t ML, float * att) {
    return attn_launch(st, false, qkv, pos_dev, nh, nkv, hd, (uint16_t *) k_cache, (uint16_t *) v_cache, ML, att, 0, 10000.0f);
}
// RoPE + KV-cache write + attention in one launch (qkv holds the UN-rotated projections)
int launch_rope_kv_attn_decode(hipStream_t st, const float * qkv, const int32_t * pos_dev, int nh, int nkv, int hd, int mode, float freq_base,
                               uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * att) {
    return attn_launch(st, true, qkv, pos_dev, nh, nkv, hd, k_cache, v_cache, ML, att, mode, freq_base);
}
extern "C" int cllm_op_attn_decode(void * stream, const float * q, const int32_t * pos_dev, int n_head, int n_kv_head, int head_dim,
                                   const void * k_cache, const void * v_cache, int64_t max_len, float * out) {
    if (!q || !pos_dev || !k_cache || !v_cache || !out || n_head <= 0 || n_kv_head <= 0 || head_dim <= 0) FAIL(CLLM_E_INVALID, "attn_decode: arguments");
    return launch_attn_decode((hipStream_t) stream, q, pos_dev, n_head, n_kv_head, head_dim, (const uint16_t *) k_cache, (const uint16_t *) v_cache, max_len, out);
}

// The whole single-token attention block between the q/k/v projections and o_proj as ONE call (1 launch up to the long-context
// threshold, 3 above it): ROPE(q), ROPE(k) + SET_ROWS(k_cache), CPY(v -> v_cache column), MUL_MAT(K,Q), SCALE, DIAG_MASK_INF,
// SOFT_MAX, MUL_MAT(V,P), PERMUTE + CONT.  qkv: the UN-rotated projections [n_head*hd | n_kv_head*hd | n_kv_head*hd] F32.
extern "C" int cllm_op_rope_table(void * stream, const int32_t * pos_dev, int head_dim, float freq_base, float * cs) {
    if (!pos_dev || !cs || head_dim <= 0 || head_dim % 2) FAIL(CLLM_E_INVALID, "rope_table: arguments");
    return launch_rope_table((hipStream_t) stream, pos_dev, head_dim, freq_base, cs);
}
extern "C" int cllm_attn_decode_supported(int n_head, int n_kv_head, int head_dim, int64_t max_len) {       // the shapes attn_launch() takes
    if (n_head <= 0 || n_kv_head <= 0 || head_dim <= 0 || max_len <= 0) return 0;
    if (head_dim % 8 || max_len % 8 || n_head % n_kv_head) return 0;
    return (size_t)(3 * head_dim + (max_len > head_dim ? max_len : head_dim)) * 4 <= 150 * 1024;
}
extern "C" size_t cllm_attn_decode_wsize(int64_t n_kv, int n_head, int64_t max_len) {
    return n_kv > attn_long_threshold() ? (size_t) n_head * (size_t) max_len * 6 : 0;
}
extern "C" int cllm_op_rope_kv_attn_decode(void * stream, const float * qkv, const int32_t * pos_dev, const float * rope_cs, float freq_base, int64_t n_kv,
                                           int n_head, int n_kv_head, int head_dim, int rope_mode, void * k_cache, void * v_cache, int64_t max_len,
                                           float * out, void * wdata, size_t wsize) {
    if (!qkv || !pos_dev || !k_cache || !v_cache || !out || n_head <= 0 || n_kv_head <= 0 || head_dim <= 0 || n_kv <= 0 || n_kv > max_len)
        FAIL(CLLM_E_INVALID, "rope_kv_attn_decode: arguments");
    if (rope_mode != 0 && rope_mode != 2) FAIL(CLLM_E_UNSUPPORTED, "rope_kv_attn_decode: rope mode %d", rope_mode);
    hipStream_t st = (hipStream_t) stream;
    int rc = CLLM_E_UNSUPPORTED;
    const size_t need = cllm_attn_decode_wsize(n_kv, n_head, max_len);
    if (rope_cs && need && wdata && wsize >= need)
        rc = launch_attn_long_flash(st, qkv, pos_dev, rope_cs, n_head, n_kv_head, head_dim, rope_mode, (uint16_t *) k_cache, (uint16_t *) v_cache, max_len, (float *) wdata, wsize, out);
    if (rc == CLLM_E_UNSUPPORTED && rope_cs && need && wdata && wsize >= need)
        rc = launch_attn_long(st, qkv, pos_dev, rope_cs, n_head, n_kv_head, head_dim, rope_mode, (uint16_t *) k_cache, (uint16_t *) v_cache, max_len, (float *) wdata, out);
    if (rc == CLLM_E_UNSUPPORTED && rope_cs)
        rc = launch_attn_dec_table(st, qkv, pos_dev, rope_cs, n_head, n_kv_head, head_dim, rope_mode, (uint16_t *) k_cache, (uint16_t *) v_cache, max_len, out);
    if (rc == CLLM_E_UNSUPPORTED)
        rc = launch_rope_kv_attn_decode(st, qkv, pos_dev, n_head, n_kv_head, head_dim, rope_mode, freq_base, (uint16_t *) k_cache, (uint16_t *) v_cache, max_len, out);
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------
// embedding gather for the token held in device memory is cllm_op_get_rows; greedy sampling + advance:
// ---------------------------------------------------------------------------------------------------------------
// greedy sampler in two short launches (a single workgroup scanning 128K logits took 43 us):
//   stage 1: 256 workgroups each reduce a contiguous slice to (value, index), first maximum wins
//   stage 2: one workgroup reduces the 256 partials in slice order and advances the device-side loop state
__device__ __forceinline__ void argmax_combine(float & best, int & idx, float ov, int oi) {
    if (ov > best || (ov == best && oi < idx)) { best = ov; idx = oi; }
}
__global__ void __launch_bounds__(256) k_argmax_partial(const float * __restrict__ x, int n, float * __restrict__ pv, int * __restrict__ pi) {
    __shared__ float bv[4]; __shared__ int bi[4];
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) { const float v = x[i]; if (v > best) { best = v; idx = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        pv[blockIdx.x] = best; pi[blockIdx.x] = idx;
    }
}
__global__ void __launch_bounds__(256) k_argmax_final(const float * __restrict__ pv, const int * __restrict__ pi, int np, int32_t * __restrict__ tok_dev,
                                                      int32_t * __restrict__ pos_dev, int32_t * __restrict__ out_ring, int32_t * __restrict__ counter) {
    __shared__ float bv[4]; __shared__ int bi[4];
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < np; i += 256) argmax_combine(best, idx, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        if (idx == 0x7fffffff) idx = 0;
        tok_dev[0] = idx;                 // next step's input token
        pos_dev[0] = pos_dev[0] + 1;      // and its position
        out_ring[counter[0]] = idx;
        counter[0] = counter[0] + 1;
    }
}
// The second stage AND the head of the NEXT step in one launch (the runner's greedy loop: one workgroup is on the chip anyway): the token's embedding row (GET_ROWS,
// dequant_elem: the bits of k_get_rows) into x_next and the cos / sin table of position pos + 1 (k_rope_table's code, by the last wave while thread 0 finishes the reduce).
// A decode step then is [layers] [lm_head + first stage] [this]: three launches fewer than get_rows, rope_table, ..., lm_head, partial, final.
__global__ void __launch_bounds__(1024) k_argmax_final_next(const float * __restrict__ pv, const int * __restrict__ pi, int np, int32_t * __restrict__ tok_dev,
                                                            int32_t * __restrict__ pos_dev, int32_t * __restrict__ out_ring, int32_t * __restrict__ counter,
                                                            int etype, const char * __restrict__ emb, size_t emb_nb1, int H, float * __restrict__ x_next,
                                                            int half, float theta_scale, float * __restrict__ cs) {
    __shared__ float bv[16]; __shared__ int bi[16]; __shared__ int tok_s;
    const int tid = threadIdx.x;
    const int pos_next = pos_dev[0] + 1;              // (read by everybody BEFORE the barrier thread 0 stores behind)
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = tid; i < np; i += 1024) argmax_combine(best, idx, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 16; w++) argmax_combine(best, idx, bv[w], bi[w]);
        if (idx == 0x7fffffff) idx = 0;
        tok_s = idx;
        tok_dev[0] = idx;                 // next step's input token
        pos_dev[0] = pos_next;            // and its position
        out_ring[counter[0]] = idx;
        counter[0] = counter[0] + 1;
    }
    if (cs && tid >= 960)
        for (int i = tid - 960; i < half; i += 64) {
            float theta = (float) pos_next;
            for (int k = 0; k < i; k++) theta *= theta_scale;        // iterated fp32 multiplication, ops.cpp:5639-5650
            float c, s_;
            rope_cos_sin(theta, &c, &s_);
            cs[2*i] = c * 1.0f; cs[2*i + 1] = s_ * 1.0f;
        }
    __syncthreads();
    const char * row = emb + (size_t) tok_s * emb_nb1;
    for (int c = tid; c < H; c += 1024) x_next[c] = dequant_elem(etype, row, c);
}
int launch_argmax_partial(hipStream_t st, const float * logits, int n, float * part_v, int * part_i) {
    hipLaunchKernelGGL(k_argmax_partial, dim3(256), dim3(256), 0, st, logits, n, part_v, part_i);
    LAUNCH_CHECK();
    return CLLM_OK;
}
// hd == 0: no cos / sin table (head sizes the compact attention kernel does not take)
int launch_argmax_final_next(hipStream_t st, const float * part_v, const int * part_i, int np, int32_t * tok_dev, int32_t * pos_dev, int32_t * out_ring, int32_t * counter,
                             int etype, const void * emb, size_t emb_nb1, int H, float * x_next, int hd, float freq_base, float * cs) {
    hipLaunchKernelGGL(k_argmax_final_next, dim3(1), dim3(1024), 0, st, part_v, part_i, np, tok_dev, pos_dev, out_ring, counter, etype, (const char *) emb, emb_nb1, H, x_next,
                       hd / 2, hd ? powf(freq_base, -2.0f / hd) : 0.0f, hd ? cs : (float *) nullptr);
    LAUNCH_CHECK();
    return CLLM_OK;
}
__global__ void __launch_bounds__(256) k_argmax_publish(const float * __restrict__ pv, const int * __restrict__ pi, int np, int32_t * __restrict__ tok_dev,
                                                        int32_t * __restrict__ tok_host, int32_t * const * __restrict__ inc, int n_inc) {
    __shared__ float bv[4]; __shared__ int bi[4];
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < np; i += 256) argmax_combine(best, idx, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        if (idx == 0x7fffffff) idx = 0;
        tok_dev[0] = idx;
        if (tok_host) { tok_host[0] = idx; __threadfence_system(); }
    }
    for (int i = threadIdx.x; i < n_inc; i += 256) inc[i][0] = inc[i][0] + 1;
}
// the same with ABSOLUTE values: every (pointer, value) record of set_table_dev[] is stored (no read-modify-write of memory the previous graph's later
// nodes may have reused)
struct argmax_set_rec { int32_t * ptr; int32_t val; int32_t pad; };
__global__ void __launch_bounds__(256) k_argmax_publish_set(const float * __restrict__ pv, const int * __restrict__ pi, int np, int32_t * __restrict__ tok_dev,
                                                            int32_t * __restrict__ tok_host, const argmax_set_rec * __restrict__ recs, int n_set) {
    __shared__ float bv[4]; __shared__ int bi[4];
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < np; i += 256) argmax_combine(best, idx, pv[i], pi[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((threadIdx.x & 63) == 0) { bv[threadIdx.x >> 6] = best; bi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        if (idx == 0x7fffffff) idx = 0;
        tok_dev[0] = idx;
        if (tok_host) { tok_host[0] = idx; __threadfence_system(); }
    }
    for (int i = threadIdx.x; i < n_set; i += 256) recs[i].ptr[0] = recs[i].val;
}
extern "C" int cllm_op_argmax_set(void * stream, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host, const void * set_table_dev, int n_set, void * scratch) {
    if (!logits || n <= 0 || n > INT32_MAX || !tok_dev || !scratch || n_set < 0 || (n_set && !set_table_dev)) FAIL(CLLM_E_INVALID, "argmax_set: arguments");
    float * pv = (float *) scratch; int * pi = (int *)((char *) scratch + 1024);
    hipLaunchKernelGGL(k_argmax_partial, dim3(256), dim3(256), 0, (hipStream_t) stream, logits, (int) n, pv, pi);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_argmax_publish_set, dim3(1), dim3(256), 0, (hipStream_t) stream, (const float *) pv, (const int *) pi, 256, tok_dev, tok_host, (const argmax_set_rec *) set_table_dev, n_set);
    LAUNCH_CHECK();
    return CLLM_OK;
}
// Snapshot + greedy sampler + scalar updates of a step started ahead of the host (host/ggml-hip.cpp ahead_launch) as ONE launch: through the unmodified host these were five
// (three device-to-device copies of the graph's outputs, k_argmax_partial, k_argmax_publish_set: 24 us of kernels + their boundaries in front of every token,
// profiles/r06_token_timeline_through_the_host.txt).  256 workgroups copy the ranges of `ranges_dev` (16-byte words where both sides are aligned, else 4-byte) and reduce their
// slice of the logits (k_argmax_partial's slices and tie rule: first maximum); the LAST workgroup to arrive (ticket) reduces the 256 partials in slice order, publishes the token
// and stores the (pointer, value) records -- every read of the logits has finished by then (the token id's block may be part of them: ggml-alloc).
struct snap_range { const char * src; char * dst; unsigned long long bytes; };
__global__ void __launch_bounds__(256) k_snapshot_argmax_set(const snap_range * __restrict__ rng, int n_rng, const float * __restrict__ x, int n, float * pv, int * pi, unsigned * ticket,
                                                             int32_t * __restrict__ tok_dev, int32_t * __restrict__ tok_host, const argmax_set_rec * __restrict__ recs, int n_set) {
    __shared__ float bv[4]; __shared__ int bi[4]; __shared__ int last_s;
    const int tid = threadIdx.x, gsz = gridDim.x * 256, gid = blockIdx.x * 256 + tid;
    for (int r = 0; r < n_rng; r++) {
        const snap_range R = rng[r];
        if ((((uintptr_t) R.src | (uintptr_t) R.dst | R.bytes) & 15) == 0) { for (unsigned long long i = gid; i < R.bytes / 16; i += gsz) ((u32x4 *) R.dst)[i] = ((const u32x4 *) R.src)[i]; }
        else                                                               { for (unsigned long long i = gid; i < R.bytes / 4; i += gsz) ((uint32_t *) R.dst)[i] = ((const uint32_t *) R.src)[i]; }
    }
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int lo = blockIdx.x * per, hi = min(n, lo + per);
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = lo + tid; i < hi; i += 256) { const float v = x[i]; if (v > best) { best = v; idx = i; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();                                               // (also: every thread's copies and logit reads are issued and their values consumed)
    if (tid == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        __hip_atomic_store(pv + blockIdx.x, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(pi + blockIdx.x, idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();                                           // this workgroup's stores (snapshot, partial) before its ticket
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        last_s = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!last_s) return;
    __threadfence();
    best = -INFINITY; idx = 0x7fffffff;
    for (int i = tid; i < (int) gridDim.x; i += 256)
        argmax_combine(best, idx, __hip_atomic_load(pv + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(pi + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) argmax_combine(best, idx, __shfl_xor(best, o, 64), __shfl_xor(idx, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) { bv[tid >> 6] = best; bi[tid >> 6] = idx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < 4; w++) argmax_combine(best, idx, bv[w], bi[w]);
        if (idx == 0x7fffffff) idx = 0;
        tok_dev[0] = idx;
        if (tok_host) { tok_host[0] = idx; __threadfence_system(); }
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next launch starts from zero
    }
    for (int i = tid; i < n_set; i += 256) recs[i].ptr[0] = recs[i].val;
}
// ranges_dev: n_ranges x { src, dst, bytes } (device table; bytes % 4 == 0); scratch: 4096 bytes of device memory, ZEROED by the caller before the first use (partials + ticket)
extern "C" __attribute__((visibility("default")))
int cllm_op_snapshot_argmax_set(void * stream, const void * ranges_dev, int n_ranges, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host,
                                const void * set_table_dev, int n_set, void * scratch) {
    if (!logits || n <= 0 || n > INT32_MAX || !tok_dev || !scratch || n_set < 0 || (n_set && !set_table_dev) || n_ranges < 0 || (n_ranges && !ranges_dev)) FAIL(CLLM_E_INVALID, "snapshot_argmax_set: arguments");
    float * pv = (float *) scratch; int * pi = (int *)((char *) scratch + 1024); unsigned * ticket = (unsigned *)((char *) scratch + 2048);
    hipLaunchKernelGGL(k_snapshot_argmax_set, dim3(256), dim3(256), 0, (hipStream_t) stream, (const snap_range *) ranges_dev, n_ranges, logits, (int) n, pv, pi, ticket, tok_dev, tok_host,
                       (const argmax_set_rec *) set_table_dev, n_set);
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" int cllm_op_argmax_advance(void * stream, const float * logits, int64_t n, int32_t * tok_dev, int32_t * tok_host, int32_t * const * inc_ptrs_dev, int n_inc,
                                      void * scratch) {
    if (!logits || n <= 0 || n > INT32_MAX || !tok_dev || !scratch || n_inc < 0 || (n_inc && !inc_ptrs_dev)) FAIL(CLLM_E_INVALID, "argmax_advance: arguments");
    float * pv = (float *) scratch; int * pi = (int *)((char *) scratch + 1024);
    hipLaunchKernelGGL(k_argmax_partial, dim3(256), dim3(256), 0, (hipStream_t) stream, logits, (int) n, pv, pi);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_argmax_publish, dim3(1), dim3(256), 0, (hipStream_t) stream, (const float *) pv, (const int *) pi, 256, tok_dev, tok_host, inc_ptrs_dev, n_inc);
    LAUNCH_CHECK();
    return CLLM_OK;
}
int launch_argmax_advance(hipStream_t st, const float * logits, int n, int32_t * tok_dev, int32_t * pos_dev, int32_t * out_ring, int32_t * counter,
                          float * part_v, int * part_i) {
    const int np = 256;
    hipLaunchKernelGGL(k_argmax_partial, dim3(np), dim3(256), 0, st, logits, n, part_v, part_i);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_argmax_final, dim3(1), dim3(256), 0, st, (const float *) part_v, (const int *) part_i, np, tok_dev, pos_dev, out_ring, counter);
    LAUNCH_CHECK();
    return CLLM_OK;
}
