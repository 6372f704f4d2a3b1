// ops.hip -- the non-matmul nodes of one chatllm.cpp forward graph, written for gfx950.
// Each kernel cites the reference CPU function whose arithmetic (operation order, rounding points,
// accumulation width) it reproduces; see include/chatllm_hip.h for the op contracts.
#include "common.h"
#include "quant_dev.h"

#include <math.h>

// ================================================================================================
// RMS_NORM (+ MUL)      ggml_compute_forward_rms_norm_f32, ggml-cpu/ops.cpp:3710-3759
//   sum of x*x in double, mean -> float, scale = 1/sqrtf(mean+eps), y = x*scale   [, y *= w as a second rounding]
// one workgroup per row; HBM/L2-bound, float4 traffic.
// ================================================================================================
template <bool MUL>
__global__ void __launch_bounds__(1024) k_rms_norm(tview s, tview d, const float * __restrict__ w, int64_t w_ne0, float eps) {
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % s.ne[1], i2 = (row / s.ne[1]) % s.ne[2], i3 = row / (s.ne[1] * s.ne[2]);
    const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
    float *       y = (float *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
    const int64_t n = s.ne[0];
    __shared__ double part[16];
    const bool aligned = (((uintptr_t) x) & 15) == 0;
    const f32x4 first = (int64_t) threadIdx.x < (n >> 2) ? rms_load4(x, threadIdx.x, aligned) : f32x4{0, 0, 0, 0};
    const double sum = rms_block_sumsq_1024(x, n, first, part);
    const float scale = rms_scale(sum, n, eps, x, nullptr, part);
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) {
        float v = x[i] * scale;
        if (MUL) v = v * w[i % w_ne0];
        y[i] = v;
    }
}

static int rms_norm_impl(void * stream, const cllm_tensor * src, const cllm_tensor * weight, cllm_tensor * dst, float eps) {
    if (!src || !dst) FAIL(CLLM_E_INVALID, "rms_norm: null");
    if (src->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32) FAIL(CLLM_E_UNSUPPORTED, "rms_norm: type");
    if (!t_same_shape(src, dst) || src->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_INVALID, "rms_norm: shape");
    if (weight && (weight->type != CLLM_TYPE_F32 || weight->nb[0] != 4 || weight->ne[0] != src->ne[0] || t_nrows(weight) != 1))
        FAIL(CLLM_E_UNSUPPORTED, "rms_norm_mul: weight must be a dense F32 [ne0] vector");
    const int64_t rows = t_nrows(src);
    if (rows == 0 || src->ne[0] == 0) return CLLM_OK;
    hipStream_t st = (hipStream_t) stream;
    if (weight) hipLaunchKernelGGL(k_rms_norm<true>,  dim3((unsigned) rows), dim3(1024), 0, st, tv(src), tv(dst), (const float *) weight->data, weight->ne[0], eps);
    else        hipLaunchKernelGGL(k_rms_norm<false>, dim3((unsigned) rows), dim3(1024), 0, st, tv(src), tv(dst), (const float *) nullptr, (int64_t) 1, eps);
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" int cllm_op_rms_norm(void * stream, const cllm_tensor * src, cllm_tensor * dst, float eps) { return rms_norm_impl(stream, src, nullptr, dst, eps); }
extern "C" int cllm_op_rms_norm_mul(void * stream, const cllm_tensor * src, const cllm_tensor * weight, cllm_tensor * dst, float eps) {
    if (!weight) FAIL(CLLM_E_INVALID, "rms_norm_mul: null weight");
    return rms_norm_impl(stream, src, weight, dst, eps);
}

// ================================================================================================
// ROPE        ggml_compute_forward_rope_flt<float>, ggml-cpu/ops.cpp:5589-5865
//   theta_i built by ITERATED fp32 multiplication (theta *= theta_scale), cos/sin per (token, pair),
//   YaRN mixing (rope_yarn :5596-5611).  One workgroup per token: the first n_dims/2 threads build
//   the cos/sin cache in LDS, then all threads rotate that token's heads.  In-place safe.
// ================================================================================================
struct rope_k_params { int n_dims, mode; float theta_scale, freq_scale, ext_factor, attn_factor, corr0, corr1, mscale_yarn; };

__global__ void __launch_bounds__(256) k_rope(tview s, tview d, const int32_t * __restrict__ pos, const float * __restrict__ ff, rope_k_params p) {
    extern __shared__ float cache[];                 // [n_dims]: cos, sin interleaved
    const int64_t i2 = blockIdx.x % s.ne[2], i3 = blockIdx.x / s.ne[2];
    const int half = p.n_dims / 2;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float theta = (float) pos[i2];
        for (int k = 0; k < i; k++) theta *= p.theta_scale;
        const float f = ff ? ff[i] : 1.0f;
        const float theta_extrap = theta / f;
        const float theta_interp = p.freq_scale * theta_extrap;
        float th = theta_interp, mscale = p.attn_factor;
        if (p.ext_factor != 0.0f) {
            const float y = ((float) i - p.corr0) / fmaxf(0.001f, p.corr1 - p.corr0);      // i == i0/2
            const float ramp = (1.0f - fminf(1.0f, fmaxf(0.0f, y))) * p.ext_factor;
            th = __builtin_fmaf(theta_interp, 1 - ramp, theta_extrap * ramp);      // (gcc contracts the reference's statement into this fma: oracle rope_yarn)
            mscale = p.mscale_yarn;                                                // attn_factor * fma(0.1f, logf(1 / freq_scale), 1.0f): the host's libm (launcher)
        }
        float cs, sn;
        rope_cos_sin(th, &cs, &sn);
        cache[2*i]     = cs * mscale;
        cache[2*i + 1] = sn * mscale;
    }
    __syncthreads();
    const int64_t ne0 = s.ne[0], nh = s.ne[1];
    const int64_t off = p.mode == 0 ? 1 : half;
    // rotated pairs
    for (int64_t t = threadIdx.x; t < nh * half; t += blockDim.x) {
        const int64_t h = t / half; const int i = (int)(t % half);
        const float * x = (const float *)(s.data + h*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        float *       y = (float *)(d.data + h*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
        const int64_t ic = p.mode == 0 ? 2*i : i;
        const float c = cache[2*i], sn = cache[2*i + 1];
        const float x0 = x[ic], x1 = x[ic + off];
        y[ic]       = rope_rot_a(x0, x1, c, sn);
        y[ic + off] = rope_rot_b(x0, x1, c, sn);
    }
    // pass-through channels
    if (p.n_dims < ne0 && s.data != d.data) {
        const int64_t rest = ne0 - p.n_dims;
        for (int64_t t = threadIdx.x; t < nh * rest; t += blockDim.x) {
            const int64_t h = t / rest, i0 = p.n_dims + t % rest;
            *(float *)(d.data + i0*4 + h*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]) = *(const float *)(s.data + i0*4 + h*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        }
    }
}

static float rope_corr_dim(int n_dims, int n_ctx_orig, float n_rot, float base) {       // ggml.c ggml_rope_yarn_corr_dim
    return n_dims * logf(n_ctx_orig / (n_rot * 2 * (float) M_PI)) / (2 * logf(base));
}

extern "C" int cllm_op_rope(void * stream, const cllm_tensor * src, const cllm_tensor * pos, const cllm_tensor * freq_factors,
                            cllm_tensor * dst, const cllm_rope_params * p) {
    if (!src || !pos || !dst || !p) FAIL(CLLM_E_INVALID, "rope: null");
    if (src->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || pos->type != CLLM_TYPE_I32) FAIL(CLLM_E_UNSUPPORTED, "rope: type");
    if (p->mode != 0 && p->mode != 2) FAIL(CLLM_E_UNSUPPORTED, "rope: mode %d", p->mode);
    if (!t_same_shape(src, dst) || src->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_INVALID, "rope: shape");
    if (p->n_dims > src->ne[0] || (p->n_dims & 1) || p->n_dims <= 0) FAIL(CLLM_E_INVALID, "rope: n_dims");
    if (pos->ne[0] < src->ne[2]) FAIL(CLLM_E_INVALID, "rope: pos shorter than sequence");
    if (freq_factors && (freq_factors->type != CLLM_TYPE_F32 || freq_factors->ne[0] < p->n_dims / 2)) FAIL(CLLM_E_INVALID, "rope: freq_factors");
    if (t_nelements(src) == 0) return CLLM_OK;
    rope_k_params k;
    k.n_dims = p->n_dims; k.mode = p->mode;
    k.theta_scale = powf(p->freq_base, -2.0f / p->n_dims);
    k.freq_scale = p->freq_scale; k.ext_factor = p->ext_factor; k.attn_factor = p->attn_factor;
    const float start = floorf(rope_corr_dim(p->n_dims, p->n_ctx_orig, p->beta_fast, p->freq_base));
    const float end   = ceilf (rope_corr_dim(p->n_dims, p->n_ctx_orig, p->beta_slow, p->freq_base));
    k.corr0 = fmaxf(0.0f, start); k.corr1 = fminf((float)(p->n_dims - 1), end);
    k.mscale_yarn = p->attn_factor * fmaf(0.1f, logf(1.0f / p->freq_scale), 1.0f);      // glibc's logf on the host: the reference's own value (ops.cpp:5607)
    hipLaunchKernelGGL(k_rope, dim3((unsigned)(src->ne[2] * src->ne[3])), dim3(256), (size_t) p->n_dims * 4, (hipStream_t) stream,
                       tv(src), tv(dst), (const int32_t *) pos->data, freq_factors ? (const float *) freq_factors->data : nullptr, k);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// The runner's prefill: ROPE(k) -> SET_ROWS(k_cache), TRANSPOSE(v) -> CPY(v_cache column), ROPE(q) of one token per workgroup in ONE launch
// (KVCacheAttention::forward src/layers.cpp:2925-2945, 3082-3093).  The cos / sin values, the rotation and the fp32 -> fp16 conversions are those of
// k_rope / k_set_rows / k_cpy above (plain RoPE: no frequency factors, freq_scale 1, no YaRN -- the cllm_llama configuration): same bits in q and in
// the caches.  qkv: [q | k | v] rows of QKV floats per token; q is rotated in place, k and v are only read.
__global__ void __launch_bounds__(256) k_rope_kv_store(float * __restrict__ qkv, int64_t QKV, const int32_t * __restrict__ pos, int nh, int nkv, int hd, int mode, float theta_scale,
                                                       uint16_t * __restrict__ k_cache, uint16_t * __restrict__ v_cache, int64_t ML) {
    extern __shared__ float cache[];                 // [hd]: cos, sin interleaved
    const int64_t tok = blockIdx.x;
    const int half = hd / 2, p = pos[tok];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        float theta = (float) p;
        for (int k = 0; k < i; k++) theta *= theta_scale;
        float cs, sn;
        rope_cos_sin(theta, &cs, &sn);
        cache[2*i] = cs; cache[2*i + 1] = sn;
    }
    __syncthreads();
    const int64_t QD = (int64_t) nh * hd, KD = (int64_t) nkv * hd;
    const int off = mode == 0 ? 1 : half;
    float * q = qkv + tok * QKV;
    const float * k = q + QD, * v = k + KD;
    for (int t = threadIdx.x; t < nh * half; t += blockDim.x) {
        const int h = t / half, i = t % half, ic = mode == 0 ? 2*i : i;
        const float c = cache[2*i], sn = cache[2*i + 1];
        float * x = q + (int64_t) h * hd;
        const float x0 = x[ic], x1 = x[ic + off];
        x[ic] = rope_rot_a(x0, x1, c, sn); x[ic + off] = rope_rot_b(x0, x1, c, sn);
    }
    if (p < 0 || p >= ML) return;                    // (the CPU asserts; never write out of bounds)
    uint16_t * kr = k_cache + (int64_t) p * KD;
    for (int t = threadIdx.x; t < nkv * half; t += blockDim.x) {
        const int h = t / half, i = t % half, ic = mode == 0 ? 2*i : i;
        const float c = cache[2*i], sn = cache[2*i + 1];
        const float * x = k + (int64_t) h * hd;
        const float x0 = x[ic], x1 = x[ic + off];
        kr[(int64_t) h * hd + ic] = f2h(rope_rot_a(x0, x1, c, sn)); kr[(int64_t) h * hd + ic + off] = f2h(rope_rot_b(x0, x1, c, sn));
    }
    for (int64_t d = threadIdx.x; d < KD; d += blockDim.x) v_cache[d * ML + p] = f2h(v[d]);
}
int launch_rope_kv_store(hipStream_t st, float * qkv, int64_t QKV, const int32_t * pos, int64_t n_tok, int nh, int nkv, int hd, int mode, float freq_base,
                         void * k_cache, void * v_cache, int64_t ML) {
    if ((mode != 0 && mode != 2) || hd <= 0 || (hd & 1) || n_tok <= 0) return CLLM_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_rope_kv_store, dim3((unsigned) n_tok), dim3(256), (size_t) hd * 4, st, qkv, QKV, pos, nh, nkv, hd, mode, powf(freq_base, -2.0f / hd),
                       (uint16_t *) k_cache, (uint16_t *) v_cache, ML);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ================================================================================================
// SOFT_MAX (+ SCALE + DIAG_MASK_INF)   ggml_compute_forward_soft_max_f32 ops.cpp:5225-5335,
//   ggml_vec_soft_max_f32 vec.cpp:547- (groups of 8 through ggml_v_expf, f32 tree hsum, total in double; expf tail)
// one wave per row; lane handles groups of 8 consecutive elements (the CPU's AVX2 vector), so the
// exponentials AND the per-group partial sums are bit-identical to the CPU's.
// ================================================================================================
template <int MODE>   // 0: plain (+scale, +mask tensor)   1: fused scale + causal mask(n_past)
__global__ void __launch_bounds__(256) k_soft_max(tview s, tview d, const char * __restrict__ mask, int mask_f16, int64_t m_nb1, int64_t m_nb2, int64_t m_nb3,
                                                  int64_t m_ne2, int64_t m_ne3, float scale, int n_past) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nrows = s.ne[1] * s.ne[2] * s.ne[3];
    if (row >= nrows) return;
    const int64_t i1 = row % s.ne[1], i2 = (row / s.ne[1]) % s.ne[2], i3 = row / (s.ne[1] * s.ne[2]);
    const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
    float *       y = (float *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
    const char * mp = mask ? mask + i1*m_nb1 + (i2 % m_ne2)*m_nb2 + (i3 % m_ne3)*m_nb3 : nullptr;
    const int64_t n = s.ne[0];

    auto val = [&](int64_t i) -> float {
        float v = x[i] * scale;
        if (MODE == 1) { if (i > (int64_t) n_past + i1) v = -INFINITY; }
        else if (mp)   { v += 1.0f * (mask_f16 ? h2f(((const uint16_t *) mp)[i]) : ((const float *) mp)[i]); }
        return v;
    };
    // MODE 1: everything from n_vis on is masked (-inf -> probability exactly 0): those entries are neither read nor
    // exponentiated, whole groups of 8 beyond it contribute an exact 0.0 to the sum, and zeros are stored for them
    const int64_t n_vis = MODE == 1 ? (n_past + i1 + 1 < n ? n_past + i1 + 1 : n) : n;
    float mx = -INFINITY;
    for (int64_t i = lane; i < n_vis; i += 64) mx = fmaxf(mx, val(i));
    mx = wave_max(mx);

    const int64_t nv = n & ~(int64_t) 7;
    const int64_t nv_vis = MODE == 1 ? (((n_vis + 7) & ~(int64_t) 7) < nv ? ((n_vis + 7) & ~(int64_t) 7) : nv) : nv;
    double sum = 0.0;
    if (MODE == 1) for (int64_t i = nv_vis + lane; i < nv; i += 64) y[i] = 0.0f;
    for (int64_t g = (int64_t) lane * 8; g < nv_vis; g += 64 * 8) {
        float e[8];
#pragma unroll
        for (int l = 0; l < 8; l++) { e[l] = ggml_expf_poly(val(g + l) - mx); y[g + l] = e[l]; }
        const float a0 = e[0] + e[4], a1 = e[1] + e[5], a2 = e[2] + e[6], a3 = e[3] + e[7];
        sum += (double)((a0 + a2) + (a1 + a3));
    }
    if (lane == 0) for (int64_t i = nv; i < n; i++) { const float e = libm_expf(val(i) - mx); y[i] = e; sum += (double) e; }
    sum = wave_sum_d(sum);
    double rinv = 1.0 / sum;                                 // (ORDER: soft_total_order_safe, common.h)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // e[] written by other lanes of this wave
    if (__builtin_expect(!soft_total_order_safe(rinv, (int)(n >> 3)), 0)) rinv = 1.0 / soft_sum_serial(y, (int) nv, (int) n);      // (masked entries hold exact zeros)
    const float inv = (float) rinv;
    const int64_t n_live = MODE == 1 ? (nv_vis < nv ? nv_vis : n) : n;      // the stored zeros stay zeros
    for (int64_t i = lane; i < n_live; i += 64) y[i] = y[i] * inv;
}

// fused scale + causal mask + soft_max with the row held in registers (prefill score rows: one read, one write instead of
// two of each).  Same lane -> group-of-8 mapping, same max set, same summation order as k_soft_max<1>: bit-identical.
// Needs n % 8 == 0, n <= 512 * NG, 16-byte aligned rows.
// OUT16: the probabilities are stored as fp16 (RNE of the float product: exactly what ggml's from_float makes of them when V.P takes them as src1), at the start of the
// row they came from -- the prompt's exact attention block (mmf_exact.hip) then reads half the bytes and converts nothing
template <int NG, bool OUT16 = false>
__global__ void __launch_bounds__(256) k_soft_max_causal_reg(tview s, tview d, float scale, int n_past) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t nrows = s.ne[1] * s.ne[2] * s.ne[3];
    if (row >= nrows) return;
    const int64_t i1 = row % s.ne[1], i2 = (row / s.ne[1]) % s.ne[2], i3 = row / (s.ne[1] * s.ne[2]);
    const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
    float *       y = (float *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
    const int n = (int) s.ne[0];
    const int n_vis = n_past + (int) i1 + 1 < n ? n_past + (int) i1 + 1 : n;       // entries >= n_vis are masked
    float e[NG][8];
    float mx = -INFINITY;
    // every group's loads are issued before the first value is used (unconditional, clamped to the row's start where the group is masked: inside the
    // `if` the compiler waited for each group's pair of loads before requesting the next -- NG memory round trips per row)
    f32x4 lo[NG], hi[NG];
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const int g0 = (lane + 64 * t) * 8, gc = g0 < n_vis ? g0 : 0;
        lo[t] = *(const f32x4 *)(x + gc); hi[t] = *(const f32x4 *)(x + gc + 4);
    }
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const int g0 = (lane + 64 * t) * 8;
        if (g0 < n_vis) {
            const float v[8] = { lo[t].x, lo[t].y, lo[t].z, lo[t].w, hi[t].x, hi[t].y, hi[t].z, hi[t].w };
#pragma unroll
            for (int l = 0; l < 8; l++) { e[t][l] = g0 + l < n_vis ? v[l] * scale : -INFINITY; mx = fmaxf(mx, e[t][l]); }
        }
    }
    mx = wave_max(mx);
    double sum = 0.0;
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const int g0 = (lane + 64 * t) * 8;
        if (g0 < n_vis) {
#pragma unroll
            for (int l = 0; l < 8; l++) e[t][l] = ggml_expf_poly(e[t][l] - mx);
            const float a0 = e[t][0] + e[t][4], a1 = e[t][1] + e[t][5], a2 = e[t][2] + e[t][6], a3 = e[t][3] + e[t][7];
            sum += (double)((a0 + a2) + (a1 + a3));
        }
    }
    sum = wave_sum_d(sum);
    double rinv = 1.0 / sum;                                 // (ORDER: soft_total_order_safe, common.h)
    if (__builtin_expect(!soft_total_order_safe(rinv, n >> 3), 0)) {        // the exponentials go to the output row first (it is rewritten below), then lane 0 adds them serially
        float * yy = (float *) y;
#pragma unroll
        for (int t = 0; t < NG; t++) {
            const int g0 = (lane + 64 * t) * 8;
            if (g0 < n) {
#pragma unroll
                for (int l = 0; l < 8; l++) yy[g0 + l] = g0 < n_vis ? e[t][l] : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
        rinv = 1.0 / soft_sum_serial(yy, n, n);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    const float inv = (float) rinv;
#pragma unroll
    for (int t = 0; t < NG; t++) {
        const int g0 = (lane + 64 * t) * 8;
        if (g0 < n) {
            f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if (g0 < n_vis) { lo = f32x4{e[t][0] * inv, e[t][1] * inv, e[t][2] * inv, e[t][3] * inv}; hi = f32x4{e[t][4] * inv, e[t][5] * inv, e[t][6] * inv, e[t][7] * inv}; }
            if constexpr (OUT16) {
                *(u32x4 *)((char *) y + (size_t) g0 * 2) = u32x4{ (uint32_t) f2h(lo.x) | ((uint32_t) f2h(lo.y) << 16), (uint32_t) f2h(lo.z) | ((uint32_t) f2h(lo.w) << 16),
                                                                  (uint32_t) f2h(hi.x) | ((uint32_t) f2h(hi.y) << 16), (uint32_t) f2h(hi.z) | ((uint32_t) f2h(hi.w) << 16) };
            } else { *(f32x4 *)(y + g0) = lo; *(f32x4 *)(y + g0 + 4) = hi; }
        }
    }
}

// the same for rows too long for registers (8192 < n <= 32768): one 256-thread workgroup per row, the row staged in LDS.
// All four waves exponentiate; the per-group float sums are accumulated by ONE wave in k_soft_max's order (lane l owns groups
// l, l + 64, ...; double; DPP tree), so the result is still bit-identical to the single-wave kernel.
__global__ void __launch_bounds__(256) k_soft_max_causal_lds(tview s, tview d, float scale, int n_past) {
    extern __shared__ __attribute__((aligned(16))) float sm[];      // [n] values | [n / 8] group sums
    __shared__ float red_f[4];
    __shared__ double red_d[1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int64_t i1 = row % s.ne[1], i2 = (row / s.ne[1]) % s.ne[2], i3 = row / (s.ne[1] * s.ne[2]);
    const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
    float *       y = (float *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
    const int n = (int) s.ne[0];
    const int n_vis = n_past + (int) i1 + 1 < n ? n_past + (int) i1 + 1 : n;
    const int nv_vis = (n_vis + 7) & ~7;                              // n % 8 == 0: whole groups
    float * gsum = sm + n;
    float mx = -INFINITY;
    for (int g0 = tid * 8; g0 < nv_vis; g0 += 256 * 8) {
        const f32x4 lo = *(const f32x4 *)(x + g0), hi = *(const f32x4 *)(x + g0 + 4);
        const float v[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
#pragma unroll
        for (int l = 0; l < 8; l++) { const float t = g0 + l < n_vis ? v[l] * scale : -INFINITY; sm[g0 + l] = t; mx = fmaxf(mx, t); }
    }
    mx = wave_max(mx);
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red_f[0], red_f[1]), fmaxf(red_f[2], red_f[3]));
    for (int g0 = tid * 8; g0 < nv_vis; g0 += 256 * 8) {              // each thread re-reads what it wrote
        float e[8];
#pragma unroll
        for (int l = 0; l < 8; l++) { e[l] = ggml_expf_poly(sm[g0 + l] - mx); sm[g0 + l] = e[l]; }
        const float a0 = e[0] + e[4], a1 = e[1] + e[5], a2 = e[2] + e[6], a3 = e[3] + e[7];
        gsum[g0 >> 3] = (a0 + a2) + (a1 + a3);
    }
    __syncthreads();
    if (wave == 0) {
        double sum = 0.0;
        for (int gq = lane; gq < (nv_vis >> 3); gq += 64) sum += (double) gsum[gq];
        sum = wave_sum_d(sum);
        double rinv = 1.0 / sum;                                      // (ORDER: soft_total_order_safe, common.h; groups beyond nv_vis are exact zeros in the reference's sum)
        if (__builtin_expect(!soft_total_order_safe(rinv, n >> 3), 0)) rinv = 1.0 / soft_sum_serial_groups(gsum, nv_vis >> 3, sm, 0, 0);
        if (lane == 0) red_d[0] = rinv;
    }
    __syncthreads();
    const float inv = (float) red_d[0];
    for (int g0 = tid * 8; g0 < n; g0 += 256 * 8) {
        f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
        if (g0 < nv_vis) { lo = f32x4{sm[g0] * inv, sm[g0 + 1] * inv, sm[g0 + 2] * inv, sm[g0 + 3] * inv}; hi = f32x4{sm[g0 + 4] * inv, sm[g0 + 5] * inv, sm[g0 + 6] * inv, sm[g0 + 7] * inv}; }
        *(f32x4 *)(y + g0) = lo; *(f32x4 *)(y + g0 + 4) = hi;
    }
}

// SCALE + DIAG_MASK_INF + SOFT_MAX over f32 rows [n, rows ...] IN PLACE, the probabilities left as fp16 at the start of each row (see k_soft_max_causal_reg<.., true>);
// CLLM_E_UNSUPPORTED: shapes the register-resident kernel does not take (the caller keeps f32)
int launch_soft_max_causal_f16out(hipStream_t st, const tview & sv, float scale, int n_past) {
    const int64_t n = sv.ne[0], rows = sv.ne[1] * sv.ne[2] * sv.ne[3];
    if (n % 8 || n < 512 || n > 8192 || rows <= 0 || (((uintptr_t) sv.data | (uintptr_t) sv.nb[1] | (uintptr_t) sv.nb[2] | (uintptr_t) sv.nb[3]) & 15) || sv.nb[0] != 4) return CLLM_E_UNSUPPORTED;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    if (n <= 4096) hipLaunchKernelGGL((k_soft_max_causal_reg<8, true>),  dim3(grid), dim3(256), 0, st, sv, sv, scale, n_past);
    else           hipLaunchKernelGGL((k_soft_max_causal_reg<16, true>), dim3(grid), dim3(256), 0, st, sv, sv, scale, n_past);
    LAUNCH_CHECK();
    return CLLM_OK;
}

static int soft_max_impl(void * stream, const cllm_tensor * src, const cllm_tensor * mask, cllm_tensor * dst, float scale, int fused, int n_past) {
    if (!src || !dst) FAIL(CLLM_E_INVALID, "soft_max: null");
    if (src->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32) FAIL(CLLM_E_UNSUPPORTED, "soft_max: type");
    if (!t_same_shape(src, dst) || src->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_INVALID, "soft_max: shape");
    if (mask) {
        if (mask->type != CLLM_TYPE_F32 && mask->type != CLLM_TYPE_F16) FAIL(CLLM_E_UNSUPPORTED, "soft_max: mask type");
        if (mask->ne[0] != src->ne[0] || mask->ne[1] < src->ne[1] || src->ne[2] % mask->ne[2] || src->ne[3] % mask->ne[3]) FAIL(CLLM_E_INVALID, "soft_max: mask shape");
    }
    const int64_t rows = t_nrows(src);
    if (rows == 0 || src->ne[0] == 0) return CLLM_OK;
    const unsigned grid = (unsigned)((rows + 3) / 4);
    hipStream_t st = (hipStream_t) stream;
    const int64_t n = src->ne[0];
    const bool al16 = !((((uintptr_t) src->data | src->nb[1] | src->nb[2] | src->nb[3] | (uintptr_t) dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3])) & 15);
    if (fused && n % 8 == 0 && n >= 512 && n <= 8192 && al16) {      // long (prefill) rows: register-resident single pass
        if (n <= 4096) hipLaunchKernelGGL(k_soft_max_causal_reg<8>,  dim3(grid), dim3(256), 0, st, tv(src), tv(dst), scale, n_past);
        else           hipLaunchKernelGGL(k_soft_max_causal_reg<16>, dim3(grid), dim3(256), 0, st, tv(src), tv(dst), scale, n_past);
        LAUNCH_CHECK();
        return CLLM_OK;
    }
    if (fused && n % 8 == 0 && n > 8192 && n <= 32768 && al16 && rows <= 0x7fffffff) {      // very long rows: one workgroup per row, row in LDS
        const size_t lds = (size_t)(n + n / 8) * 4;
        static uint64_t attr = 0;
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_soft_max_causal_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); dev_flag_set(attr); }
        hipLaunchKernelGGL(k_soft_max_causal_lds, dim3((unsigned) rows), dim3(256), lds, st, tv(src), tv(dst), scale, n_past);
        LAUNCH_CHECK();
        return CLLM_OK;
    }
    if (fused) hipLaunchKernelGGL(k_soft_max<1>, dim3(grid), dim3(256), 0, st, tv(src), tv(dst), (const char *) nullptr, 0, (int64_t) 0, (int64_t) 0, (int64_t) 0, (int64_t) 1, (int64_t) 1, scale, n_past);
    else       hipLaunchKernelGGL(k_soft_max<0>, dim3(grid), dim3(256), 0, st, tv(src), tv(dst), mask ? (const char *) mask->data : nullptr, mask && mask->type == CLLM_TYPE_F16 ? 1 : 0,
                                  mask ? (int64_t) mask->nb[1] : 0, mask ? (int64_t) mask->nb[2] : 0, mask ? (int64_t) mask->nb[3] : 0, mask ? mask->ne[2] : 1, mask ? mask->ne[3] : 1, scale, 0);
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" int cllm_op_soft_max(void * stream, const cllm_tensor * src, const cllm_tensor * mask, cllm_tensor * dst, float scale, float max_bias) {
    if (max_bias != 0.0f) FAIL(CLLM_E_UNSUPPORTED, "soft_max: ALiBi (max_bias != 0) is not on this path");
    return soft_max_impl(stream, src, mask, dst, scale, 0, 0);
}
extern "C" int cllm_op_scale_mask_soft_max(void * stream, const cllm_tensor * src, cllm_tensor * dst, float scale, int n_past) {
    if (n_past < 0) FAIL(CLLM_E_INVALID, "soft_max: n_past");
    return soft_max_impl(stream, src, nullptr, dst, scale, 1, n_past);
}

// ================================================================================================
// element-wise family: generic strided, one thread per element of dst, index decomposed over ne[]
// ================================================================================================
enum { EW_DIAG_MASK = 0, EW_SCALE = 1, EW_SILU = 2, EW_ADD = 3, EW_MUL = 4, EW_SILU_MUL = 5, EW_DIV = 6 };

template <int OP>
__global__ void __launch_bounds__(256) k_elementwise(tview a, tview b, tview d, float f0, float f1, int i0p) {
    const int64_t n = d.ne[0] * d.ne[1] * d.ne[2] * d.ne[3];
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t i0 = r % d.ne[0]; r /= d.ne[0];
        const int64_t i1 = r % d.ne[1]; r /= d.ne[1];
        const int64_t i2 = r % d.ne[2]; const int64_t i3 = r / d.ne[2];
        const float x = *(const float *)(a.data + i0*a.nb[0] + i1*a.nb[1] + i2*a.nb[2] + i3*a.nb[3]);
        float y;
        if (OP == EW_DIAG_MASK)      y = (i0 > (int64_t) i0p + i1) ? -INFINITY : x;      // ops.cpp:5137-5185
        else if (OP == EW_SCALE)     y = (f1 == 0.0f) ? x * f0 : x * f0 + f1;            // ggml_vec_scale_f32 / ggml_vec_mad1_f32
        else if (OP == EW_SILU) {
            // ggml_vec_silu_f32 (vec.cpp:396-431): AVX2 body for i0 < (ne0 & ~7), scalar expf tail
            y = (i0 < (d.ne[0] & ~(int64_t) 7)) ? x / (1.0f + ggml_expf_poly(0.0f - x)) : x / (1.0f + libm_expf(-x));
        } else {
            const float z = *(const float *)(b.data + (i0 % b.ne[0])*b.nb[0] + (i1 % b.ne[1])*b.nb[1] + (i2 % b.ne[2])*b.nb[2] + (i3 % b.ne[3])*b.nb[3]);
            if (OP == EW_ADD)      y = x + z;
            else if (OP == EW_MUL) y = x * z;
            else if (OP == EW_DIV) y = __fdiv_rn(x, z);                                     // IEEE division (ggml_vec_div_f32)
            else {                                                                          // silu(gate) * up
                const float sl = (i0 < (d.ne[0] & ~(int64_t) 7)) ? x / (1.0f + ggml_expf_poly(0.0f - x)) : x / (1.0f + libm_expf(-x));
                y = sl * z;
            }
        }
        *(float *)(d.data + i0*d.nb[0] + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]) = y;
    }
}

template <int OP>
static int ew_launch(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * d, float f0, float f1, int ip, const char * name) {
    if (!a || !d) FAIL(CLLM_E_INVALID, "%s: null", name);
    if (a->type != CLLM_TYPE_F32 || d->type != CLLM_TYPE_F32 || (b && b->type != CLLM_TYPE_F32)) FAIL(CLLM_E_UNSUPPORTED, "%s: type", name);
    if (!t_same_shape(a, d)) FAIL(CLLM_E_INVALID, "%s: shape", name);
    if (b) for (int i = 0; i < 4; i++) if (b->ne[i] <= 0 || a->ne[i] % b->ne[i]) FAIL(CLLM_E_INVALID, "%s: src1 not broadcastable", name);
    const int64_t n = t_nelements(d);
    if (n == 0) return CLLM_OK;
    int64_t grid = (n + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_elementwise<OP>, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(a), b ? tv(b) : tv(a), tv(d), f0, f1, ip);
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" int cllm_op_diag_mask_inf(void * stream, const cllm_tensor * src, cllm_tensor * dst, int n_past) {
    if (n_past < 0) FAIL(CLLM_E_INVALID, "diag_mask_inf: n_past");
    return ew_launch<EW_DIAG_MASK>(stream, src, nullptr, dst, 0, 0, n_past, "diag_mask_inf");
}
extern "C" int cllm_op_scale(void * stream, const cllm_tensor * src, cllm_tensor * dst, float s, float b) { return ew_launch<EW_SCALE>(stream, src, nullptr, dst, s, b, 0, "scale"); }
extern "C" int cllm_op_unary(void * stream, int op, const cllm_tensor * src, cllm_tensor * dst) {
    if (op != CLLM_UNARY_SILU) FAIL(CLLM_E_UNSUPPORTED, "unary: op %d", op);
    if (src && dst && (src->nb[0] != 4 || dst->nb[0] != 4)) FAIL(CLLM_E_UNSUPPORTED, "unary: rows must be dense");
    return ew_launch<EW_SILU>(stream, src, nullptr, dst, 0, 0, 0, "silu");
}
extern "C" int cllm_op_add(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst) { if (!b) FAIL(CLLM_E_INVALID, "add: null"); return ew_launch<EW_ADD>(stream, a, b, dst, 0, 0, 0, "add"); }
extern "C" int cllm_op_mul(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst) { if (!b) FAIL(CLLM_E_INVALID, "mul: null"); return ew_launch<EW_MUL>(stream, a, b, dst, 0, 0, 0, "mul"); }
extern "C" int cllm_op_div(void * stream, const cllm_tensor * a, const cllm_tensor * b, cllm_tensor * dst) { if (!b) FAIL(CLLM_E_INVALID, "div: null"); return ew_launch<EW_DIV>(stream, a, b, dst, 0, 0, 0, "div"); }
extern "C" int cllm_op_silu_mul(void * stream, const cllm_tensor * g, const cllm_tensor * u, cllm_tensor * dst) { if (!u) FAIL(CLLM_E_INVALID, "silu_mul: null"); return ew_launch<EW_SILU_MUL>(stream, g, u, dst, 0, 0, 0, "silu_mul"); }

// ================================================================================================
// SUM_ROWS / TOP_K: the router of a sparse-MoE block (GenericSparseMLP::forward, src/layers.cpp:3755-3815): rows of n_expert values
// ================================================================================================
// ggml_compute_forward_sum_rows_f32 (ops.cpp:1451-1482) -> ggml_vec_sum_f32 (vec.h:1510-1520): sequential, accumulated in double
__global__ void __launch_bounds__(256) k_sum_rows(tview s, tview d) {
    const int64_t rows = s.ne[1] * s.ne[2] * s.ne[3];
    for (int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i1 = r % s.ne[1], i2 = (r / s.ne[1]) % s.ne[2], i3 = r / (s.ne[1] * s.ne[2]);
        const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        double sum = 0.0;
        for (int64_t i = 0; i < s.ne[0]; i++) sum += (double) x[i];
        *(float *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]) = (float) sum;
    }
}
extern "C" int cllm_op_sum_rows(void * stream, const cllm_tensor * src, cllm_tensor * dst) {
    if (!src || !dst) FAIL(CLLM_E_INVALID, "sum_rows: null");
    if (src->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_F32 || src->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_UNSUPPORTED, "sum_rows: dense F32 rows");
    if (dst->ne[0] != 1 || dst->ne[1] != src->ne[1] || dst->ne[2] != src->ne[2] || dst->ne[3] != src->ne[3]) FAIL(CLLM_E_INVALID, "sum_rows: shape");
    const int64_t rows = src->ne[1] * src->ne[2] * src->ne[3];
    if (rows == 0) return CLLM_OK;
    int64_t grid = (rows + 255) / 256; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_sum_rows, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(src), tv(dst));
    LAUNCH_CHECK();
    return CLLM_OK;
}
// ggml_compute_forward_top_k_f32 (ops.cpp:8057-8094): the indices of the k largest values in descending order (std::partial_sort with
// data[a] > data[b]), then the first two swapped ("the order is not important").  Equal values: the lower index first (the reference
// leaves that to its heap; softmax probabilities of distinct logits do not tie).  One thread per row: rows are n_expert long.
__global__ void __launch_bounds__(256) k_top_k(tview s, tview d, int k) {
    const int64_t rows = s.ne[1] * s.ne[2] * s.ne[3];
    const int n = (int) s.ne[0];
    for (int64_t r = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (int64_t) gridDim.x * blockDim.x) {
        const int64_t i1 = r % s.ne[1], i2 = (r / s.ne[1]) % s.ne[2], i3 = r / (s.ne[1] * s.ne[2]);
        const float * x = (const float *)(s.data + i1*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        int32_t * out = (int32_t *)(d.data + i1*d.nb[1] + i2*d.nb[2] + i3*d.nb[3]);
        top_k_row(x, n, k, out);
    }
}
extern "C" int cllm_op_top_k(void * stream, const cllm_tensor * src, cllm_tensor * dst) {
    if (!src || !dst) FAIL(CLLM_E_INVALID, "top_k: null");
    if (src->type != CLLM_TYPE_F32 || dst->type != CLLM_TYPE_I32 || src->nb[0] != 4 || dst->nb[0] != 4) FAIL(CLLM_E_UNSUPPORTED, "top_k: F32 rows -> I32 rows");
    const int64_t k = dst->ne[0];
    if (k < 1 || k > src->ne[0] || dst->ne[1] != src->ne[1] || dst->ne[2] != src->ne[2] || dst->ne[3] != src->ne[3]) FAIL(CLLM_E_INVALID, "top_k: shape");
    if (src->ne[0] > 4096) FAIL(CLLM_E_UNSUPPORTED, "top_k: rows of %lld values", (long long) src->ne[0]);
    const int64_t rows = src->ne[1] * src->ne[2] * src->ne[3];
    if (rows == 0) return CLLM_OK;
    int64_t grid = (rows + 255) / 256; if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_top_k, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(src), tv(dst), (int) k);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// The tail of GenericSparseMLP::forward / forward_with_experts (src/layers.cpp:3792-3872) as one launch:
//   weights = GET_ROWS(probs, ids); weights /= SUM_ROWS(weights)  [norm_topk_prob]; experts *= weights; out = experts[:,0] + experts[:,1] + ... (+ resid)
// Same operations in the same order as the nodes (double-accumulated sum, IEEE division, left-to-right adds): bit-identical.
__global__ void __launch_bounds__(256) k_moe_combine(tview e, tview p, tview ids, tview r, tview d, int k, int has_resid) {
    const int64_t H = d.ne[0], n = d.ne[0] * d.ne[1];
    for (int64_t x = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; x < n; x += (int64_t) gridDim.x * blockDim.x) {
        const int64_t h = x % H, t = x / H;
        double s = 0.0;
        for (int j = 0; j < k; j++) {
            const int32_t id = *(const int32_t *)(ids.data + j*ids.nb[0] + t*ids.nb[1]);
            s += (double) *(const float *)(p.data + (int64_t) id*4 + t*p.nb[1]);
        }
        const float sum = (float) s;
        float acc = 0.0f;
        for (int j = 0; j < k; j++) {
            const int32_t id = *(const int32_t *)(ids.data + j*ids.nb[0] + t*ids.nb[1]);
            const float w = __fdiv_rn(*(const float *)(p.data + (int64_t) id*4 + t*p.nb[1]), sum);
            const float y = *(const float *)(e.data + h*4 + j*e.nb[1] + t*e.nb[2]) * w;
            acc = j == 0 ? y : acc + y;
        }
        if (has_resid) acc = acc + *(const float *)(r.data + h*4 + t*r.nb[1]);
        *(float *)(d.data + h*4 + t*d.nb[1]) = acc;
    }
}
extern "C" int cllm_op_moe_combine(void * stream, const cllm_tensor * experts, const cllm_tensor * probs, const cllm_tensor * ids, const cllm_tensor * resid,
                                   cllm_tensor * dst) {
    if (!experts || !probs || !ids || !dst) FAIL(CLLM_E_INVALID, "moe_combine: null");
    if (experts->type != CLLM_TYPE_F32 || probs->type != CLLM_TYPE_F32 || ids->type != CLLM_TYPE_I32 || dst->type != CLLM_TYPE_F32 || (resid && resid->type != CLLM_TYPE_F32))
        FAIL(CLLM_E_UNSUPPORTED, "moe_combine: types");
    const int64_t H = experts->ne[0], k = experts->ne[1], T = experts->ne[2];
    if (k < 1 || k > 64 || ids->ne[0] != k || ids->ne[1] != T || probs->ne[1] != T || dst->ne[0] != H || dst->ne[1] != T || experts->ne[3] != 1 ||
        experts->nb[0] != 4 || probs->nb[0] != 4 || dst->nb[0] != 4 || (resid && (resid->ne[0] != H || resid->ne[1] != T || resid->nb[0] != 4)))
        FAIL(CLLM_E_INVALID, "moe_combine: shapes");
    if (H * T == 0) return CLLM_OK;
    int64_t grid = (H * T + 255) / 256; if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_moe_combine, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(experts), tv(probs), tv(ids), resid ? tv(resid) : tv(dst), tv(dst), (int) k, resid ? 1 : 0);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ================================================================================================
// SET_ROWS (K-cache write)   ggml_compute_forward_set_rows_f32, ops.cpp:4892-4940
// ================================================================================================
template <typename IDX, bool F16>
__global__ void __launch_bounds__(256) k_set_rows(tview s, tview idx, tview d) {
    const int64_t n = s.ne[0] * s.ne[1] * s.ne[2] * s.ne[3];
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t c  = r % s.ne[0]; r /= s.ne[0];
        const int64_t i  = r % s.ne[1]; r /= s.ne[1];
        const int64_t i2 = r % s.ne[2]; const int64_t i3 = r / s.ne[2];
        const int64_t row = (int64_t) *(const IDX *)(idx.data + i*idx.nb[0] + (i2 % idx.ne[1])*idx.nb[1] + (i3 % idx.ne[2])*idx.nb[2]);
        if (row < 0 || row >= d.ne[1]) continue;     // the CPU asserts; never write out of bounds
        const float v = *(const float *)(s.data + c*4 + i*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        char * dp = d.data + row*d.nb[1] + i2*d.nb[2] + i3*d.nb[3];
        if (F16) ((uint16_t *) dp)[c] = f2h(v); else ((float *) dp)[c] = v;
    }
}
// Q8_0 rows (--cache_dtype q8_0, src/layers.cpp:2925-2945): from_float of the CPU traits = quantize_row_q8_0 (x86 branch), 8 lanes per 32-block
template <typename IDX>
__global__ void __launch_bounds__(256) k_set_rows_q8_0(tview s, tview idx, tview d) {
    const int64_t nb = s.ne[0] / 32, nblk = nb * s.ne[1] * s.ne[2] * s.ne[3];
    const int sub = threadIdx.x & 7;
    for (int64_t g0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) >> 3; g0 - ((threadIdx.x & 63) >> 3) < nblk; g0 += ((int64_t) gridDim.x * blockDim.x) >> 3) {
        const bool live = g0 < nblk;                            // whole waves stay in the loop: the 8-lane reductions read their neighbours by DPP
        int64_t r = live ? g0 : nblk - 1;
        const int64_t blk = r % nb; r /= nb;
        const int64_t i  = r % s.ne[1]; r /= s.ne[1];
        const int64_t i2 = r % s.ne[2]; const int64_t i3 = r / s.ne[2];
        const int64_t row = (int64_t) *(const IDX *)(idx.data + i*idx.nb[0] + (i2 % idx.ne[1])*idx.nb[1] + (i3 % idx.ne[2])*idx.nb[2]);
        const f32x4 v = *(const f32x4 *)(s.data + (blk * 32 + sub * 4) * 4 + i*s.nb[1] + i2*s.nb[2] + i3*s.nb[3]);
        float dd; int ss;
        const uint32_t p = quant4_q8_0<false>(v, &dd, &ss);
        if (!live || row < 0 || row >= d.ne[1]) continue;
        char * dp = d.data + row*d.nb[1] + i2*d.nb[2] + i3*d.nb[3] + blk * 34;
        if (sub == 0) *(uint16_t *) dp = f2h(dd);
        *(uint16_t *)(dp + 2 + sub * 4) = (uint16_t) p; *(uint16_t *)(dp + 4 + sub * 4) = (uint16_t)(p >> 16);
    }
}
extern "C" int cllm_op_set_rows(void * stream, const cllm_tensor * src, const cllm_tensor * idx, cllm_tensor * dst) {
    if (!src || !idx || !dst) FAIL(CLLM_E_INVALID, "set_rows: null");
    if (src->type == CLLM_TYPE_F32 && dst->type == CLLM_TYPE_Q8_0) {
        if (idx->type != CLLM_TYPE_I32 && idx->type != CLLM_TYPE_I64) FAIL(CLLM_E_UNSUPPORTED, "set_rows: index type");
        if (dst->ne[0] != src->ne[0] || dst->ne[2] != src->ne[2] || dst->ne[3] != src->ne[3] || src->nb[0] != 4 || dst->nb[0] != 34 || src->ne[0] % 32) FAIL(CLLM_E_INVALID, "set_rows: shape");
        if (idx->ne[0] != src->ne[1] || src->ne[2] % idx->ne[1] || src->ne[3] % idx->ne[2]) FAIL(CLLM_E_INVALID, "set_rows: index shape");
        if ((((uintptr_t) src->data | src->nb[1] | src->nb[2] | src->nb[3]) & 15) || (((uintptr_t) dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 1)) FAIL(CLLM_E_UNSUPPORTED, "set_rows: alignment");
        const int64_t nblk = t_nelements(src) / 32;
        if (nblk == 0) return CLLM_OK;
        int64_t grid = (nblk * 8 + 255) / 256; if (grid > 8192) grid = 8192;
        if (idx->type == CLLM_TYPE_I64) hipLaunchKernelGGL((k_set_rows_q8_0<int64_t>), dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(src), tv(idx), tv(dst));
        else                            hipLaunchKernelGGL((k_set_rows_q8_0<int32_t>), dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(src), tv(idx), tv(dst));
        LAUNCH_CHECK();
        return CLLM_OK;
    }
    if (src->type != CLLM_TYPE_F32 || (dst->type != CLLM_TYPE_F16 && dst->type != CLLM_TYPE_F32)) FAIL(CLLM_E_UNSUPPORTED, "set_rows: type");
    if (idx->type != CLLM_TYPE_I32 && idx->type != CLLM_TYPE_I64) FAIL(CLLM_E_UNSUPPORTED, "set_rows: index type");
    if (dst->ne[0] != src->ne[0] || dst->ne[2] != src->ne[2] || dst->ne[3] != src->ne[3] || src->nb[0] != 4 || dst->nb[0] != cllm_type_size(dst->type)) FAIL(CLLM_E_INVALID, "set_rows: shape");
    if (idx->ne[0] != src->ne[1] || src->ne[2] % idx->ne[1] || src->ne[3] % idx->ne[2]) FAIL(CLLM_E_INVALID, "set_rows: index shape");
    const int64_t n = t_nelements(src);
    if (n == 0) return CLLM_OK;
    int64_t grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
    hipStream_t st = (hipStream_t) stream;
    const bool f16 = dst->type == CLLM_TYPE_F16, i64 = idx->type == CLLM_TYPE_I64;
    if (f16 && !i64)       hipLaunchKernelGGL((k_set_rows<int32_t, true>),  dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(idx), tv(dst));
    else if (f16 && i64)   hipLaunchKernelGGL((k_set_rows<int64_t, true>),  dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(idx), tv(dst));
    else if (!f16 && !i64) hipLaunchKernelGGL((k_set_rows<int32_t, false>), dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(idx), tv(dst));
    else                   hipLaunchKernelGGL((k_set_rows<int64_t, false>), dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(idx), tv(dst));
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ================================================================================================
// CPY / DUP / CONT    ggml_compute_forward_dup, ops.cpp:47-330,526: element e of src (row-major over ne) -> element e of dst
// ================================================================================================
template <typename TS, typename TD>
__global__ void __launch_bounds__(256) k_cpy(tview s, tview d) {
    const int64_t n = s.ne[0] * s.ne[1] * s.ne[2] * s.ne[3];
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t s0 = r % s.ne[0]; r /= s.ne[0];
        const int64_t s1 = r % s.ne[1]; r /= s.ne[1];
        const int64_t s2 = r % s.ne[2]; const int64_t s3 = r / s.ne[2];
        r = e;
        const int64_t d0 = r % d.ne[0]; r /= d.ne[0];
        const int64_t d1 = r % d.ne[1]; r /= d.ne[1];
        const int64_t d2 = r % d.ne[2]; const int64_t d3 = r / d.ne[2];
        const TS v = *(const TS *)(s.data + s0*s.nb[0] + s1*s.nb[1] + s2*s.nb[2] + s3*s.nb[3]);
        TD o;
        if constexpr (sizeof(TS) == sizeof(TD)) o = (TD) v;
        else if constexpr (sizeof(TS) == 4) o = f2h(v);          // F32 -> F16, RNE
        else o = h2f(v);                                         // F16 -> F32
        *(TD *)(d.data + d0*d.nb[0] + d1*d.nb[1] + d2*d.nb[2] + d3*d.nb[3]) = o;
    }
}
// same quantized type on both sides (CONT of the permuted Q8_0 K-cache view, src/layers.cpp:3144-3146): whole rows move, 2 bytes per thread step
__global__ void __launch_bounds__(256) k_cpy_qrows(tview s, tview d, int row_bytes) {
    const int64_t h = row_bytes / 2, n = h * s.ne[1] * s.ne[2] * s.ne[3];
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t c = r % h; r /= h;
        const int64_t row = r;
        const int64_t s1 = r % s.ne[1]; r /= s.ne[1];
        const int64_t s2 = r % s.ne[2]; const int64_t s3 = r / s.ne[2];
        r = row;
        const int64_t d1 = r % d.ne[1]; r /= d.ne[1];
        const int64_t d2 = r % d.ne[2]; const int64_t d3 = r / d.ne[2];
        *(uint16_t *)(d.data + c*2 + d1*d.nb[1] + d2*d.nb[2] + d3*d.nb[3]) = *(const uint16_t *)(s.data + c*2 + s1*s.nb[1] + s2*s.nb[2] + s3*s.nb[3]);
    }
}
extern "C" int cllm_op_cpy(void * stream, const cllm_tensor * src, cllm_tensor * dst) {
    if (!src || !dst) FAIL(CLLM_E_INVALID, "cpy: null");
    if (src->type == dst->type && (is_quant_type(src->type) || is_kq_type(src->type))) {
        const size_t rb = cllm_row_size(src->type, src->ne[0]);
        if (src->ne[0] != dst->ne[0] || t_nelements(src) != t_nelements(dst) || src->nb[0] != cllm_type_size(src->type) || dst->nb[0] != src->nb[0] || rb % 2) FAIL(CLLM_E_UNSUPPORTED, "cpy: quantized rows must stay whole");
        if ((((uintptr_t) src->data | src->nb[1] | src->nb[2] | src->nb[3] | (uintptr_t) dst->data | dst->nb[1] | dst->nb[2] | dst->nb[3]) & 1)) FAIL(CLLM_E_UNSUPPORTED, "cpy: alignment");
        const int64_t n = (int64_t)(rb / 2) * src->ne[1] * src->ne[2] * src->ne[3];
        if (n == 0) return CLLM_OK;
        int64_t grid = (n + 255) / 256; if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(k_cpy_qrows, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, tv(src), tv(dst), (int) rb);
        LAUNCH_CHECK();
        return CLLM_OK;
    }
    const bool s32 = src->type == CLLM_TYPE_F32 || src->type == CLLM_TYPE_I32, s16 = src->type == CLLM_TYPE_F16;
    const bool d32 = dst->type == CLLM_TYPE_F32 || dst->type == CLLM_TYPE_I32, d16 = dst->type == CLLM_TYPE_F16;
    if (!(s32 || s16) || !(d32 || d16)) FAIL(CLLM_E_UNSUPPORTED, "cpy: type %d -> %d", src->type, dst->type);
    if ((src->type == CLLM_TYPE_I32) != (dst->type == CLLM_TYPE_I32)) FAIL(CLLM_E_UNSUPPORTED, "cpy: I32 only to I32");
    if (t_nelements(src) != t_nelements(dst)) FAIL(CLLM_E_INVALID, "cpy: element count");
    const int64_t n = t_nelements(src);
    if (n == 0) return CLLM_OK;
    int64_t grid = (n + 255) / 256; if (grid > 16384) grid = 16384;
    hipStream_t st = (hipStream_t) stream;
    if (s32 && d32)      hipLaunchKernelGGL((k_cpy<uint32_t, uint32_t>), dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(dst));
    else if (s16 && d16) hipLaunchKernelGGL((k_cpy<uint16_t, uint16_t>), dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(dst));
    else if (s32 && d16) hipLaunchKernelGGL((k_cpy<float, uint16_t>),    dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(dst));
    else                 hipLaunchKernelGGL((k_cpy<uint16_t, float>),    dim3((unsigned) grid), dim3(256), 0, st, tv(src), tv(dst));
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ================================================================================================
// GET_ROWS / dequantize    ops.cpp:4653-4700,4820 ; dequantize_row_* ggml-quants.c:307-325,401-414,1352-1373
// ================================================================================================
#include "dequant.h"
__global__ void __launch_bounds__(256) k_get_rows(int type, tview s, tview idx, tview d) {
    const int64_t n = d.ne[0] * d.ne[1] * d.ne[2] * d.ne[3];
    for (int64_t e = (int64_t) blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t) gridDim.x * blockDim.x) {
        int64_t r = e;
        const int64_t c   = r % d.ne[0]; r /= d.ne[0];
        const int64_t i10 = r % d.ne[1]; r /= d.ne[1];
        const int64_t i11 = r % d.ne[2]; const int64_t i12 = r / d.ne[2];
        const int64_t row = *(const int32_t *)(idx.data + i10*idx.nb[0] + i11*idx.nb[1] + i12*idx.nb[2]);
        float v = 0.0f;
        if (row >= 0 && row < s.ne[1]) v = dequant_elem(type, s.data + row*s.nb[1] + i11*s.nb[2] + i12*s.nb[3], c);
        *(float *)(d.data + c*4 + i10*d.nb[1] + i11*d.nb[2] + i12*d.nb[3]) = v;
    }
}
extern "C" int cllm_op_get_rows(void * stream, const cllm_tensor * src, const cllm_tensor * idx, cllm_tensor * dst) {
    if (!src || !idx || !dst) FAIL(CLLM_E_INVALID, "get_rows: null");
    if (idx->type != CLLM_TYPE_I32 || dst->type != CLLM_TYPE_F32) FAIL(CLLM_E_UNSUPPORTED, "get_rows: type");
    switch (src->type) { case CLLM_TYPE_F32: case CLLM_TYPE_F16: case CLLM_TYPE_Q4_0: case CLLM_TYPE_Q4_1: case CLLM_TYPE_Q8_0: case CLLM_TYPE_Q4_K: case CLLM_TYPE_Q5_K: case CLLM_TYPE_Q6_K:
        case CLLM_TYPE_Q5_0: case CLLM_TYPE_Q5_1: case CLLM_TYPE_Q2_K: case CLLM_TYPE_Q3_K: case CLLM_TYPE_IQ4_NL: case CLLM_TYPE_MXFP4: case CLLM_TYPE_IQ4_XS: case CLLM_TYPE_TQ1_0: case CLLM_TYPE_TQ2_0:
        case CLLM_TYPE_IQ2_XXS: case CLLM_TYPE_IQ2_XS: case CLLM_TYPE_IQ2_S: case CLLM_TYPE_IQ3_XXS: case CLLM_TYPE_IQ3_S: case CLLM_TYPE_IQ1_S: case CLLM_TYPE_IQ1_M: break;
        default: FAIL(CLLM_E_UNSUPPORTED, "get_rows: src type %d", src->type); }
    if (dst->ne[0] != src->ne[0] || dst->ne[1] != idx->ne[0] || dst->ne[2] != idx->ne[1] || dst->ne[3] != idx->ne[2] || dst->nb[0] != 4) FAIL(CLLM_E_INVALID, "get_rows: shape");
    const int64_t n = t_nelements(dst);
    if (n == 0) return CLLM_OK;
    int64_t grid = (n + 255) / 256; if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_get_rows, dim3((unsigned) grid), dim3(256), 0, (hipStream_t) stream, src->type, tv(src), tv(idx), tv(dst));
    LAUNCH_CHECK();
    return CLLM_OK;
}
extern "C" int cllm_dequantize_row(void * stream, int type, const void * blocks, float * y, int64_t k) {
    if (!blocks || !y || k <= 0) FAIL(CLLM_E_INVALID, "dequantize_row: args");
    const int bs = cllm_blck_size(type);
    if (bs == 0 || k % bs) FAIL(CLLM_E_INVALID, "dequantize_row: k");
    // a 1-row table gathered with index 0
    static thread_local int32_t * zero_idx = nullptr;
    if (!zero_idx) { HIP_TRY(hipMalloc((void **) &zero_idx, 4)); HIP_TRY(hipMemset(zero_idx, 0, 4)); }
    cllm_tensor s = { type, {k, 1, 1, 1}, {cllm_type_size(type), cllm_row_size(type, k), cllm_row_size(type, k), cllm_row_size(type, k)}, (void *) blocks };
    cllm_tensor i = { CLLM_TYPE_I32, {1, 1, 1, 1}, {4, 4, 4, 4}, zero_idx };
    cllm_tensor d = { CLLM_TYPE_F32, {k, 1, 1, 1}, {4, (size_t) k*4, (size_t) k*4, (size_t) k*4}, y };
    return cllm_op_get_rows(stream, &s, &i, &d);
}
