// common.h -- shared device/host helpers for the gfx950 kernel library (not a public header).
// Formats follow /root/reference/ggml/src/ggml-common.h:170-175 (Q4_0), 219-224 (Q8_0), 295-306 (Q4_K).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/chatllm_hip.h"
#include "glibc_math.h"

#define CLLM_WAVE 64

// ---- error plumbing (host) ------------------------------------------------------------------
void cllm_set_error(const char * fmt, ...);
int  cllm_hip_check(hipError_t e, const char * what, const char * file, int line);
#define HIP_TRY(expr) do { int _rc = cllm_hip_check((expr), #expr, __FILE__, __LINE__); if (_rc) return _rc; } while (0)
#define LAUNCH_CHECK() HIP_TRY(hipGetLastError())
#define FAIL(code, ...) do { cllm_set_error(__VA_ARGS__); return (code); } while (0)

// ---- quant block formats ----------------------------------------------------------------------
#define QK    32
#define QK_K  256
struct __attribute__((packed)) block_q4_0 { uint16_t d; uint8_t qs[16]; };                       // 18 B
struct __attribute__((packed)) block_q8_0 { uint16_t d; int8_t  qs[32]; };                       // 34 B
struct __attribute__((packed)) block_q4_1 { uint16_t d, m; uint8_t qs[16]; };                    // 20 B: w = nib * d + m
struct __attribute__((packed)) block_q8_1 { uint16_t d, s; int8_t  qs[32]; };                    // 36 B: s = d * sum(qs)
struct __attribute__((packed)) block_q4_K { uint16_t d, dmin; uint8_t scales[12]; uint8_t qs[128]; }; // 144 B
struct __attribute__((packed)) block_q8_K { float d; int8_t qs[256]; int16_t bsums[16]; };      // 292 B
struct __attribute__((packed)) block_q5_K { uint16_t d, dmin; uint8_t scales[12]; uint8_t qh[32]; uint8_t qs[128]; };   // 176 B (ggml-common.h:308-321)
struct __attribute__((packed)) block_q6_K { uint8_t ql[128]; uint8_t qh[64]; int8_t scales[16]; uint16_t d; };          // 210 B (ggml-common.h:323-336)
static_assert(sizeof(block_q5_K) == 176 && sizeof(block_q6_K) == 210, "block sizes");
// the other formats stock model files carry (ggml-common.h:190-216, 262-288, 415-419): mat-mul through gemv_kq.hip, GET_ROWS
struct __attribute__((packed)) block_q5_0   { uint16_t d; uint8_t qh[4]; uint8_t qs[16]; };                  // 22 B: w = ((nib | bit << 4) - 16) * d
struct __attribute__((packed)) block_q5_1   { uint16_t d, m; uint8_t qh[4]; uint8_t qs[16]; };               // 24 B: w = (nib | bit << 4) * d + m
struct __attribute__((packed)) block_iq4_nl { uint16_t d; uint8_t qs[16]; };                                 // 18 B: w = kvalues_iq4nl[nib] * d
struct __attribute__((packed)) block_iq4_xs { uint16_t d, scales_h; uint8_t scales_l[4]; uint8_t qs[128]; };  // 136 B: 256 weights, w = kvalues_iq4nl[nib] * d * (ls - 32), a 6-bit ls per 32 (ggml-common.h:421-427)
struct __attribute__((packed)) block_tq1_0  { uint8_t qs[48]; uint8_t qh[4]; uint16_t d; };                   // 54 B: 256 ternary weights, 5 per byte of qs (base 3), 4 per byte of qh: w = (trit - 1) * d (ggml-common.h:241-249)
struct __attribute__((packed)) block_tq2_0  { uint8_t qs[64]; uint16_t d; };                                  // 66 B: 256 ternary weights, 2 bits each: w = (q - 1) * d (ggml-common.h:251-256)
static_assert(sizeof(block_tq1_0) == 54 && sizeof(block_tq2_0) == 66, "block sizes");
struct __attribute__((packed)) block_mxfp4  { uint8_t e; uint8_t qs[16]; };                                  // 17 B: w = kvalues_mxfp4[nib] * 2^(e - 128)
struct __attribute__((packed)) block_q2_K   { uint8_t scales[16]; uint8_t qs[64]; uint16_t d, dmin; };       // 84 B
struct __attribute__((packed)) block_q3_K   { uint8_t hmask[32]; uint8_t qs[64]; uint8_t scales[12]; uint16_t d; };   // 110 B
static_assert(sizeof(block_q5_0) == 22 && sizeof(block_q5_1) == 24 && sizeof(block_iq4_nl) == 18 && sizeof(block_mxfp4) == 17 && sizeof(block_q2_K) == 84 && sizeof(block_q3_K) == 110 && sizeof(block_iq4_xs) == 136, "block sizes");
static_assert(sizeof(block_q4_1) == 20 && sizeof(block_q8_1) == 36, "block sizes");
static_assert(sizeof(block_q4_0) == 18 && sizeof(block_q8_0) == 34 && sizeof(block_q4_K) == 144 && sizeof(block_q8_K) == 292, "block sizes");

// ---- device-side activation layout ("act row") -------------------------------------------------
// The CPU path converts each src1 row to the weight's vec_dot_type before the dot products
// (ggml-cpu.c:1241-1326).  We do the same on the device but keep the result split in planes so a
// wave can fetch 16-byte aligned pieces:
//   Q8_0 kind (for Q4_0/Q8_0 weights):  qs[K] int8 | d[K/32] f32 (= fp16-rounded scale) | s[K/32] int32 (sum of qs)
//   Q8_K kind (for Q4_K weights):       qs[K] int8 | d[K/256] f32                      | s[K/32] int32 (sum per 32)
//   Q8_1 kind (for Q4_1 weights):       the Q8_0 planes, but s[K/32] is an f32: fp16(d_unrounded * sum of qs), block_q8_1::s
// Each plane starts 16-byte aligned; act_row_bytes() is the per-row stride.  A "kind" is the quantization block size (32 / 256);
// ACT_Q8_1 names the third flavour and maps to block size 32.
enum { ACT_Q8_0 = 32, ACT_Q8_K = 256, ACT_Q8_1 = 33 };
__host__ __device__ inline int    act_blk(int kind) { return kind == ACT_Q8_1 ? 32 : kind; }
__host__ __device__ inline size_t act_align16(size_t x) { return (x + 15) & ~(size_t) 15; }
__host__ __device__ inline size_t act_off_d(int64_t K) { return act_align16((size_t) K); }
__host__ __device__ inline size_t act_off_s(int64_t K, int kind) { return act_off_d(K) + act_align16((size_t)(K / act_blk(kind)) * 4); }
__host__ __device__ inline size_t act_row_bytes(int64_t K, int kind) { return act_off_s(K, kind) + act_align16((size_t)(K / 32) * 4); }
// the activation format a weight type's dot product reads (type_traits_cpu[].vec_dot_type, ggml-cpu/ggml-cpu.c:207-390)
// the codebook ("grid") formats IQ2_XXS / IQ2_XS / IQ2_S / IQ3_XXS / IQ3_S (ggml-common.h:346-390; block sizes 66 / 74 / 82 / 98 / 110 bytes, fp16 d first; codebooks: iq_grids.h)
__host__ __device__ inline bool   is_iq_grid_type(int t) { return t == CLLM_TYPE_IQ2_XXS || t == CLLM_TYPE_IQ2_XS || t == CLLM_TYPE_IQ2_S || t == CLLM_TYPE_IQ3_XXS || t == CLLM_TYPE_IQ3_S; }
// the coverage types (gemv_kq.hip: mat-mul for any number of columns in the reference's order, GET_ROWS; no fused decode forms)
__host__ __device__ inline bool   is_kq_type(int t) {
    return t == CLLM_TYPE_Q5_K || t == CLLM_TYPE_Q6_K || t == CLLM_TYPE_Q2_K || t == CLLM_TYPE_Q3_K || t == CLLM_TYPE_Q5_0 || t == CLLM_TYPE_Q5_1 || t == CLLM_TYPE_IQ4_NL || t == CLLM_TYPE_MXFP4 || t == CLLM_TYPE_IQ4_XS || t == CLLM_TYPE_TQ1_0 || t == CLLM_TYPE_TQ2_0 || is_iq_grid_type(t) || t == CLLM_TYPE_IQ1_S || t == CLLM_TYPE_IQ1_M;
}
__host__ __device__ inline bool   is_k256_type(int t) { return t == CLLM_TYPE_Q4_K || t == CLLM_TYPE_Q5_K || t == CLLM_TYPE_Q6_K || t == CLLM_TYPE_Q2_K || t == CLLM_TYPE_Q3_K || t == CLLM_TYPE_IQ4_XS || t == CLLM_TYPE_TQ1_0 || t == CLLM_TYPE_TQ2_0 || is_iq_grid_type(t) || t == CLLM_TYPE_IQ1_S || t == CLLM_TYPE_IQ1_M; }
__host__ __device__ inline int    act_kind_of(int wtype) { return is_k256_type(wtype) ? ACT_Q8_K : (wtype == CLLM_TYPE_Q4_1 || wtype == CLLM_TYPE_Q5_1) ? ACT_Q8_1 : ACT_Q8_0; }
__host__ __device__ inline bool   is_quant_type(int t) { return t == CLLM_TYPE_Q4_0 || t == CLLM_TYPE_Q4_1 || t == CLLM_TYPE_Q8_0 || t == CLLM_TYPE_Q4_K; }

// ---- small device helpers -----------------------------------------------------------------------
#ifdef __HIPCC__
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float    f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float    h2f(uint16_t h) { _Float16 v; __builtin_memcpy(&v, &h, 2); return (float) v; }
// f32 -> f16, round-to-nearest-even of the f32 VALUE.  The empty asm pins the operand in a VGPR first: without it the
// compiler may fold a preceding multiply/fma into v_fma_mixlo_f16 (ONE rounding, straight to fp16), whereas the CPU path
// rounds the product to f32 and then to fp16 (two roundings) -- they differ on ties (observed: 1 soft-max probability in ~10^4).
__device__ __forceinline__ uint16_t f2h(float f)    { asm volatile("" : "+v"(f)); _Float16 v = (_Float16) f; uint16_t h; __builtin_memcpy(&h, &v, 2); return h; }

// ---- cross-lane reductions on DPP (no LDS crossbar: a ds_bpermute costs ~100 cycles, a DPP move ~8) ------------------------
// xor-1 / xor-2 inside quads, then mirror inside 8 and 16 lanes: after the four steps every lane of a 16-lane row holds the
// row total; the four row totals are combined through SGPRs (v_readlane).  All call sites share these helpers, so the fp32
// summation order is the same everywhere (fused and node-by-node paths stay bit-identical).
#define DPP_QUAD_XOR1   0xB1    // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2   0x4E    // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141   // lane i <-> 7 - i inside each 8 lanes
#define DPP_ROW_MIRROR  0x140   // lane i <-> 15 - i inside each 16 lanes
template <int CTRL> __device__ __forceinline__ int   dpp_i(int v)   { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = dpp_i<CTRL>((int) b), hi = dpp_i<CTRL>((int)(b >> 32));
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}
template <int CTRL> __device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const int lo = dpp_i<CTRL>((int) v), hi = dpp_i<CTRL>((int)(v >> 32));
    return ((unsigned long long)(unsigned int) hi << 32) | (unsigned int) lo;
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ double lane_d(double v, int l) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readlane((int) b, l), hi = __builtin_amdgcn_readlane((int)(b >> 32), l);
    return __longlong_as_double(((long long) hi << 32) | (unsigned int) lo);
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_QUAD_XOR1>(v); v += dpp_f<DPP_QUAD_XOR2>(v); v += dpp_f<DPP_HALF_MIRROR>(v); v += dpp_f<DPP_ROW_MIRROR>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v)); v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
__device__ __forceinline__ double wave_sum_d(double v) {
    v += dpp_d<DPP_QUAD_XOR1>(v); v += dpp_d<DPP_QUAD_XOR2>(v); v += dpp_d<DPP_HALF_MIRROR>(v); v += dpp_d<DPP_ROW_MIRROR>(v);
    return (lane_d(v, 0) + lane_d(v, 16)) + (lane_d(v, 32) + lane_d(v, 48));
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_i<DPP_QUAD_XOR1>(v); v += dpp_i<DPP_QUAD_XOR2>(v); v += dpp_i<DPP_HALF_MIRROR>(v); v += dpp_i<DPP_ROW_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
    unsigned long long o;
    o = dpp_u64<DPP_QUAD_XOR1>(v);   v = o > v ? o : v;
    o = dpp_u64<DPP_QUAD_XOR2>(v);   v = o > v ? o : v;
    o = dpp_u64<DPP_HALF_MIRROR>(v); v = o > v ? o : v;
    o = dpp_u64<DPP_ROW_MIRROR>(v);  v = o > v ? o : v;
    unsigned long long r[4];
#pragma unroll
    for (int i = 0; i < 4; i++) r[i] = ((unsigned long long)(unsigned int) __builtin_amdgcn_readlane((int)(v >> 32), 16 * i) << 32) | (unsigned int) __builtin_amdgcn_readlane((int) v, 16 * i);
    const unsigned long long a = r[0] > r[1] ? r[0] : r[1], b = r[2] > r[3] ? r[2] : r[3];
    return a > b ? a : b;
}
// RMS_NORM's sum of squares over one row, by a 1024-thread workgroup: thread t owns the 4-element groups t, t+1024, ...
// (accumulated in double in increasing index like the CPU loop, ops.cpp:3727-3730), a DPP wave reduction, then the 16 wave
// partials in wave order.  ONE definition shared by k_rms_norm and the fused GEMV prologue keeps the two bit-identical.
// `first` = the caller's already-loaded group t (ignored when t >= n/4); `part` = 16 doubles of LDS.  Contains a barrier.
__device__ __forceinline__ f32x4 rms_load4(const float * __restrict__ x, int64_t q, bool aligned) {
    if (aligned) return *(const f32x4 *)(x + 4 * q);
    return f32x4{ x[4*q], x[4*q + 1], x[4*q + 2], x[4*q + 3] };
}
__device__ __forceinline__ double rms_block_sumsq_1024(const float * __restrict__ x, int64_t n, f32x4 first, double * part) {
    const int tid = threadIdx.x;
    const int64_t nq = n >> 2;
    const bool aligned = (((uintptr_t) x) & 15) == 0;
    double sum = 0.0;
    f32x4 v = first;
    for (int64_t q = tid; q < nq; q += 1024) {
        if (q != tid) v = rms_load4(x, q, aligned);
        sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w);
    }
    if (4 * nq + tid < n) { const float t = x[4 * nq + tid]; sum += (double)(t * t); }
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) part[tid >> 6] = sum;
    __syncthreads();
    double tot = part[0];
#pragma unroll
    for (int w = 1; w < 16; w++) tot += part[w];
    return tot;
}

// scale = 1 / sqrt(sum / n + eps) exactly as ops.cpp:3731-3736 (double division, float sqrt, float division).  For a power
// of two n the division is an exact exponent shift: multiply by 1/n instead (same bits, ~25 instructions less on the
// decode kernels' cold critical path); the general division sits in a noinline function off the straight-line code.
//
// ORDER.  The reference adds the n squares serially in index order (ops.cpp:3736-3739); the workgroup adds them as a tree.  Two double sums of the same n
// non-negative terms differ by at most 2 gamma_(n-1) = 2 (n - 1) u / (1 - (n - 1) u) of their value (u = 2^-53; Higham, Accuracy and Stability, 4.2), and the
// only thing the rest of the op sees of the sum is (float)(sum / n): division and conversion are monotonic, so whenever the two ends of
// sum (1 -+ (2 n + 16) u) give the SAME float, the serial sum gives it too and the tree's result is the reference's, bit for bit.  When they do not (the mean
// lies within ~1e-12 of a float rounding boundary: about 3 rows in 10^5 at n = 4096) wave 0 redoes the sum in the reference's own order (rms_serial_sumsq below).  The branch
// is uniform over the workgroup (every thread holds the same `sum`); x (+ add: the tensor-parallel partial folded into the residual) must still hold the
// row, which is why the barriers sit here: no thread of an in-place launch stores before the serial pass has read.
static __device__ __noinline__ double rms_div(double sum, double n) { return sum / n; }
// (the reference's order, by wave 0: its 64 lanes load 64 consecutive values at once -- one register, the next batch requested before this one is added -- and square
//  them; the serial chain then takes them lane by lane through v_readlane, every lane computing the same sum.  As a called function with a 64-value batch in one
//  thread's registers this cost the 128-register decode kernels 4-17 spilled registers around the call.  COHERENT: a launch that is handed x inside the
//  kernel -- L1-bypassing loads; everywhere else the row was written by an earlier launch.)
template <bool COHERENT>
static __device__ __noinline__ double rms_serial_sumsq(const float * x, const float * add, int64_t n) {
    const int lane = threadIdx.x & 63;
    auto ld = [&](int64_t i) -> float {                     // element i (0 beyond the row), with the folded partial
        if (i >= n) return 0.0f;
        float v = COHERENT ? __uint_as_float(__hip_atomic_load((const unsigned *)(x + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : x[i];
        if (add) v = v + (COHERENT ? __uint_as_float(__hip_atomic_load((const unsigned *)(add + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : add[i]);
        return v;
    };
    double sum = 0.0;
    float cur = ld(lane);
    for (int64_t i = 0; i < n; i += 64) {
        const float nxt = ld(i + 64 + lane);
        const float sq = cur * cur;
        const int sqb = (int) __float_as_uint(sq);
        if (n - i >= 64) {
#pragma unroll 8
            for (int k = 0; k < 64; k++) sum += (double) __uint_as_float((unsigned) __builtin_amdgcn_readlane(sqb, k));
        } else {
            for (int k = 0; k < (int)(n - i); k++) sum += (double) __uint_as_float((unsigned) __builtin_amdgcn_readlane(sqb, k));
        }
        cur = nxt;
    }
    return sum;
}
__device__ __forceinline__ float rms_mean(double sum, int64_t n) {
    return (float)((n & (n - 1)) == 0 ? ldexp(sum, -(int) __builtin_ctzll((unsigned long long) n)) : rms_div(sum, (double) n));
}
template <bool COHERENT = false>
__device__ __forceinline__ float rms_scale(double sum, int64_t n, float eps, const float * x, const float * add, double * part) {
    float m = rms_mean(sum, n);
    const double d = sum * ((double)(2 * n + 16) * 0x1p-53);
    if (!(rms_mean(sum - d, n) == rms_mean(sum + d, n))) {
        __syncthreads();
        if (threadIdx.x < 64) { const double ss = rms_serial_sumsq<COHERENT>(x, add, n); if (threadIdx.x == 0) part[0] = ss; }
        __syncthreads();
        m = rms_mean(part[0], n);
    }
    return 1.0f / sqrtf(m + eps);
}

// the same sum for rows of at most 4096 elements (n % 4 == 0): every thread owns at most one group, no loop, no tail --
// identical bits, a fraction of the code (decode kernels: code before the main loop is fetched cold)
__device__ __forceinline__ double rms_block_sumsq_1024_one(f32x4 v, bool has, double * part) {
    const int tid = threadIdx.x;
    double sum = 0.0;
    if (has) { sum += (double)(v.x * v.x); sum += (double)(v.y * v.y); sum += (double)(v.z * v.z); sum += (double)(v.w * v.w); }
    sum = wave_sum_d(sum);
    if ((tid & 63) == 0) part[tid >> 6] = sum;
    __syncthreads();
    double tot = part[0];
#pragma unroll
    for (int w = 1; w < 16; w++) tot += part[w];
    return tot;
}

// workgroup barrier for data exchanged through LDS only: __syncthreads() also drains vmcnt, i.e. waits for every global load in flight --
// which is exactly what a latency-bound kernel prefetches ACROSS its phases.  LDS operations of a wave complete in order: lgkmcnt(0) + s_barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// load through the scalar cache (p must be wave-uniform; the data must not have been written by this kernel before).
// Scalar loads have their own counter (lgkmcnt), so they do not serialise against outstanding vector-memory prefetches.
__device__ __forceinline__ float uniform_load_f32(const float * p) {
    float v;
    asm volatile("s_load_dword %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

// reductions inside groups of 8 consecutive lanes (one 32-element quant block = 8 lanes x 4 values)
__device__ __forceinline__ float group8_max(float v) { v = fmaxf(v, dpp_f<DPP_QUAD_XOR1>(v)); v = fmaxf(v, dpp_f<DPP_QUAD_XOR2>(v)); return fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v)); }
__device__ __forceinline__ int   group8_sum_i(int v) { v += dpp_i<DPP_QUAD_XOR1>(v); v += dpp_i<DPP_QUAD_XOR2>(v); return v + dpp_i<DPP_HALF_MIRROR>(v); }
__device__ __forceinline__ int dot4(uint32_t a, uint32_t b, int c) { return __builtin_amdgcn_sdot4((int) a, (int) b, c, false); }
// Eight dot products with a zero accumulator as eight instructions: the builtin becomes v_mov_b32 0 + v_dot4c_i32_i8 (two-address form) each; the
// VOP3P form takes the inline constant.  The dot pipeline's results must not be read by another VALU instruction for 3 wait states
// (the compiler's hazard recognizer knows that for the builtin but not inside inline asm: a lone asm dot followed by a multiply returns
// garbage), so the eight go out back to back with one s_nop behind the last (tools/micro/dot4_test.hip checks the instruction itself).
__device__ __forceinline__ void dot4z_x8(const uint32_t (&a)[8], const uint32_t (&b)[8], int (&r)[8]) {
    asm("v_dot4_i32_i8 %0, %8, %16, 0\n\tv_dot4_i32_i8 %1, %9, %17, 0\n\tv_dot4_i32_i8 %2, %10, %18, 0\n\tv_dot4_i32_i8 %3, %11, %19, 0\n\t"
        "v_dot4_i32_i8 %4, %12, %20, 0\n\tv_dot4_i32_i8 %5, %13, %21, 0\n\tv_dot4_i32_i8 %6, %14, %22, 0\n\tv_dot4_i32_i8 %7, %15, %23, 0\n\ts_nop 2"
        : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]), "=&v"(r[4]), "=&v"(r[5]), "=&v"(r[6]), "=&v"(r[7])
        : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
          "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
}

// cos/sin of a RoPE angle.  ONE definition shared by k_rope (ops.hip) and k_rope_kv (decode_fused.hip): whether the
// compiler pairs cosf+sinf into a sincos or not changes the last bit for some angles, and a 1-ulp difference in q/k
// can flip their fp16 rounding -- the fused and node-by-node paths must agree to the bit.
// The CPU path calls the host's libm here (glibc 2.35): glibc_math.h reproduces its expf / sinf / cosf bit for bit (checked against the
// libm of this image over 6e8 inputs, oracle/glibc_math_check.c), where the device's own math library is only ~1 ulp close.
static __device__ __noinline__ float libm_expf(float x) { return gm_expf(x); }     // scalar tail of soft_max / SiLU: one shared body
static __device__ __noinline__ void rope_cos_sin(float theta, float * c, float * s) {
    *c = gm_cosf(theta);
    *s = gm_sinf(theta);
}

// rotate_pairs (ggml-cpu/ops.cpp:5700-5718) as the reference build compiles it: gcc contracts each expression into ONE fma around the
// rounded x1 product (determined bit for bit against libggml-cpu.so; the oracle states the same)
__device__ __forceinline__ float rope_rot_a(float x0, float x1, float c, float s) { return __builtin_fmaf(x0, c, -(x1 * s)); }
__device__ __forceinline__ float rope_rot_b(float x0, float x1, float c, float s) { return __builtin_fmaf(x0, s, x1 * c); }

// one lane of the reference's AVX2 ggml_v_expf (ggml-cpu/vec.h:1230-1267), op for op, so that
// soft_max / SiLU agree with the CPU path to the last bit wherever the CPU takes its vector body.
__device__ __forceinline__ float ggml_expf_poly(float x) {
    const float r = 0x1.8p23f;
    const float z = __builtin_fmaf(x, 0x1.715476p+0f, r);
    const float n = z - r;
    const float b = __builtin_fmaf(-n, 0x1.7f7d1cp-20f, __builtin_fmaf(-n, 0x1.62e4p-1f, x));
    const uint32_t e = __float_as_uint(z) << 23;
    const float k = __uint_as_float(e + 0x3f800000u);
    const bool  c = fabsf(n) > 126.0f;
    const float u = b * b;
    const float j = __builtin_fmaf(__builtin_fmaf(__builtin_fmaf(0x1.0e4020p-7f, b, 0x1.573e2ep-5f), u,
                                                  __builtin_fmaf(0x1.555e66p-3f, b, 0x1.fffdb6p-2f)),
                                   u, 0x1.ffffecp-1f * b);
    if (!c) return __builtin_fmaf(j, k, k);
    const uint32_t g  = (n <= 0.0f) ? 0x82000000u : 0u;
    const float    s1 = __uint_as_float(g + 0x7f000000u);
    const float    s2 = __uint_as_float(e - g);
    if (fabsf(n) > 192.0f) return s1 * s1;
    return __builtin_fmaf(s2, j, s2) * s1;
}
// ---- the soft_max's double total: ORDER.  ggml_vec_soft_max_f32 (vec.cpp:547-) adds the float sums of the groups of 8 one by one into a double, then the n mod 8
// leftovers; the kernels here add them as a tree (per-lane partial sums, a DPP tree).  Two double sums of the same m positive terms differ by at most (m + log2 m) 2^-53 of
// their value, and so do their reciprocals (+ one rounding of the division): the float that (float)(1.0 / sum) rounds to can depend on the order only if the reciprocal lies
// within that distance of a float rounding boundary -- the low 29 bits of its mantissa within (m + 24) of 2^28 (about one row in 2^20).  Every soft_max kernel tests this per
// row (as rms_scale does for RMS_NORM) and otherwise redoes the total serially in the reference's order.  SOFT_FORCE_SERIAL=1 (test builds): always serial.
#ifndef SOFT_FORCE_SERIAL
#define SOFT_FORCE_SERIAL 0
#endif
__device__ __forceinline__ bool soft_total_order_safe(double rinv, int m_terms) {
    const int low = (int)((unsigned long long) __double_as_longlong(rinv) & 0x1fffffffull) - 0x10000000;
    return !SOFT_FORCE_SERIAL && (low < 0 ? -low : low) > m_terms + 24;
}
// the total in the reference's order from the exponentials e[0 .. n) (any address space): lane 0 of the calling wave, the result broadcast
static __device__ __noinline__ double soft_sum_serial(const float * e, int nv, int n) {
    double sum = 0.0;
    if ((threadIdx.x & 63) == 0) {
        for (int gi = 0; gi < nv; gi += 8) {
            const float a0 = e[gi] + e[gi + 4], a1 = e[gi + 1] + e[gi + 5], a2 = e[gi + 2] + e[gi + 6], a3 = e[gi + 3] + e[gi + 7];
            sum += (double)((a0 + a2) + (a1 + a3));
        }
        for (int i = nv; i < n; i++) sum += (double) e[i];
    }
    return lane_d(sum, 0);
}
// the same from the group sums gs[0 .. ng) + the leftovers e[nv .. n)
static __device__ __noinline__ double soft_sum_serial_groups(const float * gs, int ng, const float * e, int nv, int n) {
    double sum = 0.0;
    if ((threadIdx.x & 63) == 0) {
        for (int gq = 0; gq < ng; gq++) sum += (double) gs[gq];
        for (int i = nv; i < n; i++) sum += (double) e[i];
    }
    return lane_d(sum, 0);
}

// SiLU as ggml_vec_silu_f32 computes it (vec.cpp:396-431): the AVX2 polynomial for the elements below n & ~7 of a row, expf for the leftovers
// SOFT_MAX of one row of n <= 512 values held in LDS by ONE wave (no scale, no mask): the lane -> group-of-8 partition, the exponentials and the
// summation order of k_soft_max (ops.hip), which follow ggml_vec_soft_max_f32 (vec.cpp:547-).  y (LDS) receives the probabilities.
__device__ __forceinline__ void wave_soft_max_plain(const float * x, float * y, int n, int lane) {
    float mx = -INFINITY;
    for (int i = lane; i < n; i += 64) mx = fmaxf(mx, x[i]);
    mx = wave_max(mx);
    const int nv = n & ~7;
    double sum = 0.0;
    for (int g = lane * 8; g < nv; g += 64 * 8) {
        float e[8];
#pragma unroll
        for (int l = 0; l < 8; l++) { e[l] = ggml_expf_poly(x[g + l] - mx); y[g + l] = e[l]; }
        const float a0 = e[0] + e[4], a1 = e[1] + e[5], a2 = e[2] + e[6], a3 = e[3] + e[7];
        sum += (double)((a0 + a2) + (a1 + a3));
    }
    if (lane == 0) for (int i = nv; i < n; i++) { const float e = libm_expf(x[i] - mx); y[i] = e; sum += (double) e; }
    sum = wave_sum_d(sum);
    double rinv = 1.0 / sum;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (__builtin_expect(!soft_total_order_safe(rinv, n >> 3), 0)) rinv = 1.0 / soft_sum_serial(y, nv, n);
    const float inv = (float) rinv;
    for (int i = lane; i < n; i += 64) y[i] = y[i] * inv;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
}
// TOP_K of one row (ggml_compute_forward_top_k_f32 ops.cpp:8057-8094): the indices of the k largest values in descending order, the first two
// swapped; equal values: the lower index first.  One thread.
__device__ __forceinline__ void top_k_row(const float * x, int n, int k, int32_t * out) {
    float prev_v = INFINITY; int prev_i = -1;                       // the last pick: the next one comes strictly after it in (value desc, index asc)
    for (int j = 0; j < k; j++) {
        float best = -INFINITY; int bi = -1;
        for (int i = 0; i < n; i++) {
            const float v = x[i];
            const bool after = v < prev_v || (v == prev_v && i > prev_i);
            if (after && (bi < 0 || v > best)) { best = v; bi = i; }
        }
        if (bi < 0) bi = 0;                                           // NaNs: the CPU's comparator is not a strict weak order there either
        out[j] = bi; prev_v = best; prev_i = bi;
    }
    if (k > 1) { const int32_t t = out[0]; out[0] = out[1]; out[1] = t; }
}
__device__ __forceinline__ float silu_poly(float x) { return x / (1.0f + ggml_expf_poly(0.0f - x)); }
__device__ __forceinline__ float silu_any(float x, bool body) { return body ? silu_poly(x) : x / (1.0f + libm_expf(-x)); }
// ---- workgroup -> output tile of a prefill GEMM, L2-aware.  A 1-D grid of TM * TN workgroups (TM token tiles, TN weight-row tiles).  The dispatcher places workgroup
// b on XCD b % 8 (private 4 MB L2s): every XCD gets one CONTIGUOUS chunk of the logical tile order (bijective for any count), and the logical order walks a group
// of GM token tiles x all row tiles with the token tile fastest, so the ~64 workgroups an XCD runs concurrently share GM activation tiles (each re-read from HBM
// once per 64 / GM row tiles instead of once per row tile) and 64 / GM weight tiles.  Placement-dependent for speed only.
__device__ __forceinline__ void gemm_tile_of(unsigned id, unsigned TM, unsigned TN, unsigned GM, unsigned & mt, unsigned & nt) {
    const unsigned nwg = TM * TN, q = nwg >> 3, r = nwg & 7, xcd = id & 7, k = id >> 3;
    const unsigned l = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;            // logical index: XCD-contiguous
    const unsigned full = TM / GM, per_full = GM * TN;
    if (l < full * per_full) { const unsigned mg = l / per_full, rr = l % per_full; nt = rr / GM; mt = GM * mg + rr % GM; }
    else { const unsigned rr = l - full * per_full, rem = TM - GM * full; nt = rr / rem; mt = GM * full + rr % rem; }
}
#endif  // __HIPCC__

// ---- host helpers ---------------------------------------------------------------------------------
static inline int64_t t_nelements(const cllm_tensor * t) { return t->ne[0]*t->ne[1]*t->ne[2]*t->ne[3]; }
static inline int64_t t_nrows(const cllm_tensor * t) { return t->ne[1]*t->ne[2]*t->ne[3]; }
static inline bool t_is_contiguous(const cllm_tensor * t) {
    size_t nb = cllm_type_size(t->type);
    if (t->nb[0] != nb) return false;
    nb *= (size_t)(t->ne[0] / cllm_blck_size(t->type));
    for (int i = 1; i < 4; i++) { if (t->ne[i] != 1 && t->nb[i] != nb) return false; nb *= (size_t) t->ne[i]; }
    return true;
}
static inline bool t_same_shape(const cllm_tensor * a, const cllm_tensor * b) {
    return a->ne[0] == b->ne[0] && a->ne[1] == b->ne[1] && a->ne[2] == b->ne[2] && a->ne[3] == b->ne[3];
}

// POD copy of a tensor's geometry for kernels
struct tview {
    char *  data;
    int64_t ne[4];
    int64_t nb[4];
};
static inline tview tv(const cllm_tensor * t) {
    tview v; v.data = (char *) t->data;
    for (int i = 0; i < 4; i++) { v.ne[i] = t->ne[i]; v.nb[i] = (int64_t) t->nb[i]; }
    return v;
}

// internal launchers (implemented across the .hip files)
int launch_quantize_act(hipStream_t st, int kind /* ACT_Q8_0 | ACT_Q8_K | ACT_Q8_1 */, const tview & src1, void * act, size_t act_stride);
// the same over silu(gate) * up, src1 rows holding 2 K interleaved (gate_e, up_e) pairs (the runner's packed gate/up projection): UNARY(SILU) + MUL + quantize in one pass
int launch_quantize_act_silu(hipStream_t st, int kind, const tview & gu, void * act, size_t act_stride);
int mmq_min_cols_get();       // capi.hip: columns from which MUL_MAT runs on the matrix cores
int launch_mmvq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, int64_t ncols_total,
                const tview & src1_geom, const tview & dst);
int launch_mmvq_id(hipStream_t st, int wtype, const tview & as, const void * act, size_t act_stride, int64_t b_ne1, const tview & ids, const tview & dst);
int launch_mmq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, const tview & src1_geom, const tview & dst, const float * resid = nullptr, int64_t ldr = 0, int epi = 0);
int launch_mul_mat_f(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d, int causal = 0, int n_past = 0);   // causal: mma_f16.hip
int launch_mma_f16(hipStream_t st, const tview & w, const tview & x, const tview & d, int causal, int n_past);
int device_cu_count();
// "done once PER DEVICE" flags (a bit per device id): hipFuncSetAttribute applies to the current device's copy of a kernel, and a host with several devices in one
// process (chatllm.cpp's layer split over this module's devices) launches every kernel on each of them
static inline bool dev_flag_unset(const uint64_t & m) { int d = 0; (void) hipGetDevice(&d); return !((m >> (d & 63)) & 1ull); }
static inline void dev_flag_set(uint64_t & m) { int d = 0; (void) hipGetDevice(&d); m |= 1ull << (d & 63); }
// per-device scratch: a buffer allocated on one device must not serve launches on another (index = device id & 63)
#define CLLM_DEV_SLOTS 64
static inline int dev_slot() { int d = 0; (void) hipGetDevice(&d); return d & (CLLM_DEV_SLOTS - 1); }
// library-owned scratch (capi.hip): one block per (device, stream, kind).  Growing it never frees a block a launch may still hold: a captured launch list (the
// host module replays hipGraphs whose signature covers only the public arguments) keeps the address it was captured with, so outgrown blocks stay alive until their
// stream is destroyed; keyed by stream, two backend contexts of one device never share one.  nullptr: allocation failed (error set).
enum { SCRATCH_ATTN_SCORES = 0, SCRATCH_X16 = 1 };
void * stream_scratch(hipStream_t st, int kind, size_t need);
void stream_scratch_release(hipStream_t st);
int launch_mmvq_act(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, const void * act, float * dst, const float * bias, const float * resid);
int launch_gemv_decode(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid, const float * padd = nullptr, float * xout = nullptr);
bool prefill_f16_enabled();
// how MUL_MAT with more than 32 activation columns and the prompt's attention block are computed (capi.hip; CLLM_PREFILL=exact | fast | f16, cllm_set_prefill_mode):
//   1 (default) exact: mmx.hip + mmf_exact.hip, the reference's accumulation order -- bit-identical to the CPU for every prompt length
//   0 fast: int8-MFMA GEMM (mmq.hip) + flash kernel (fattn.hip), their own fp32 summation order (tolerance tier)
int prefill_mode();
int prefill_attn_mode();          // the same for the prompt's attention block alone (CLLM_PREFILL_ATTN / cllm_set_prefill_attn_mode; default: follows prefill_mode())
int launch_mmx(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, const tview & src1_geom, const tview & dst, const float * resid = nullptr, int64_t ldr = 0, int epi = 0);
int launch_mmf_exact(hipStream_t st, const tview & w, const tview & x, const tview & d, int causal, int n_past, bool x_f16 = false);
int launch_soft_max_causal_f16out(hipStream_t st, const tview & sv, float scale, int n_past);      // ops.hip
// the eager attention block of a prompt in the reference's order: K.Q (causal) -> SCALE + DIAG_MASK_INF + SOFT_MAX -> V.P; q [hd, qlen, nh] F32, k [hd, n_kv, nkv] F16,
// vt [n_kv, hd, nkv] F16 (V^T rows); dst element (d, q, h) at d * 4 + q * nbn + h * nbh
int attn_prefill_exact(hipStream_t st, const tview & q, const tview & k, const tview & vt, char * dst, int64_t nbn, int64_t nbh, float scale, int n_past);
// fattn.hip: flash attention (tolerance tier).  vl 0: V rows by position [D, n_kv, ..]; vl 1: V^T rows [n_kv, D, ..] (the default V cache)
int launch_attn_long_flash(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode,
                           uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * S, size_t s_bytes, float * att);      // fattn.hip; CLLM_E_UNSUPPORTED: use launch_attn_long
int flash_prefill_min_cols();          // query rows from which the eager attention block runs as the flash kernel (CLLM_FLASH_PREFILL=0: never)
size_t fattn_wsize(int64_t N, int64_t H, int64_t B, int64_t D);
int launch_fattn(hipStream_t st, const tview & q, const tview & k, int ktype, const tview & v, int vl, const tview * mask, int causal_past,
                 char * dst, int64_t nbn, int64_t nbh, int64_t nbb, float scale, void * wdata, size_t wsize);
int launch_gemv_decode_id_combine(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, int64_t px_slot_stride,
                                  const int32_t * ids, const float * probs, const float * resid, float * dst);
int launch_quantize_act_silu2(hipStream_t st, int kind, const tview & g, const tview & u, void * act, size_t act_stride);
int launch_quantize_act_norm(hipStream_t st, int kind, const tview & s, const float * norm_w, float eps, void * act, size_t act_stride);
int launch_rope_kv_store(hipStream_t st, float * qkv, int64_t QKV, const int32_t * pos, int64_t n_tok, int nh, int nkv, int hd, int mode, float freq_base,
                         void * k_cache, void * v_cache, int64_t ML);
int launch_moe_router(hipStream_t st, int wtype, const void * W, int64_t K, int64_t n, const float * px, const float * pw, float eps,
                      float * xnorm, float * probs, int32_t * ids, int k);
int launch_gemv_decode_id_router_silu(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, const float * pw, float eps,
                                      const void * Wr, int ne, int k, float * probs, int32_t * ids, float * dst, int64_t dst_slot_stride);      // gemv_moe.hip
int launch_gemv_kq(hipStream_t st, int wtype, const tview & w, const void * act, size_t act_stride, int64_t M, float * dst, int64_t ldd);
int launch_gemv_kq_id(hipStream_t st, int wtype, const tview & as, const void * act, size_t act_stride, int64_t ne11, const tview & ids, const tview & dst);
int launch_dense_f16(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d, const float * resid = nullptr, int64_t ldr = 0, int epi = 0);
int decode_free_order();                          // gemv_free32.hip: 1 = the opt-in free-order tier of the 32-weight block formats' decode mat-vec
int launch_gemv_decode_free(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid);
int launch_gemv_rows(hipStream_t st, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid);
// ffn_fused.hip: the decode step's FFN block (norm + gate/up + SiLU*up + down + residual) as ONE launch; CLLM_E_UNSUPPORTED: the caller issues the two launches
size_t ffn_fused_state_bytes(int64_t F);
int launch_ffn_fused(hipStream_t st, const void * Wgu, const void * Wd, int64_t H, int64_t F, const float * x, const float * norm_w, float eps, void * state, bool * state_ready, float * xout);
int ffn_fused_mode();
int kernel_error_word(unsigned ** dev_ptr);      // gemv_team32.hip: the device's mapped error word (bounded in-kernel waits report there)
int launch_gemv_decode_tp_scatter(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const void * ctx_dev, int site);      // gemv_tp.hip
int launch_gemv_decode_tp_gather(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, const float * px, const float * pw, float eps, int epi, float * dst,
                                 const float * bias, const void * ctx_dev, int site, float * xout);
int launch_gemv_rows32(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid);
int gemv_team32_check();
int launch_gemv_team32(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid);
int launch_mmvq_fused(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst, const float * bias, const float * resid);
int launch_attn_decode(hipStream_t st, const float * qkv, const int32_t * pos_dev, int nh, int nkv, int hd, const uint16_t * k_cache, const uint16_t * v_cache, int64_t ML, float * att);
int launch_rope_table(hipStream_t st, const int32_t * pos_dev, int hd, float freq_base, float * cs);
int launch_attn_dec_table(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode, uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * att);
int launch_attn_long(hipStream_t st, const float * qkv, const int32_t * pos_dev, const float * rope_cs, int nh, int nkv, int hd, int mode, uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * S, float * att);
int launch_gemv_decode_id(hipStream_t st, int wtype, const void * W, size_t w_expert_bytes, int64_t K, int64_t nrows, const float * px, int64_t px_slot_stride,
                          const int32_t * ids, int n_slots, float * dst, int64_t dst_slot_stride, int epi = 0);
int attn_long_threshold();      // decoder.hip: cached positions above which the split (3-launch) attention is used
int launch_rope_kv_attn_decode(hipStream_t st, const float * qkv, const int32_t * pos_dev, int nh, int nkv, int hd, int mode, float freq_base, uint16_t * k_cache, uint16_t * v_cache, int64_t ML, float * att);
int launch_argmax_partial(hipStream_t st, const float * logits, int n, float * part_v, int * part_i);
int launch_argmax_final_next(hipStream_t st, const float * part_v, const int * part_i, int np, int32_t * tok_dev, int32_t * pos_dev, int32_t * out_ring, int32_t * counter,
                             int etype, const void * emb, size_t emb_nb1, int H, float * x_next, int hd, float freq_base, float * cs);
int launch_gemv_rows_argmax(hipStream_t st, const void * W, int64_t K, int64_t nrows, const float * px, const float * pw, float eps, float * dst, float * part_v, int * part_i, int * np);
int launch_argmax_advance(hipStream_t st, const float * logits, int n, int32_t * tok_dev, int32_t * pos_dev, int32_t * out_ring, int32_t * counter, float * part_v, int * part_i);
