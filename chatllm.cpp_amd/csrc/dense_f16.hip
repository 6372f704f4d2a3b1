// dense_f16.hip -- OPT-IN prefill contraction on the fp16 matrix cores with the block dequantization INSIDE the GEMM (north star: "per-wavefront block
// dequant staged through LDS ... MFMA tiles only for the dense bf16/fp16 prefill contraction"; SURVEY 7.1 step 5 path B; what the reference's own CUDA
// backend does above its mmq limits, ggml-cuda.cu:1229, 2262-2265: dequantize the weights, convert the activations, dense GEMM).
//
//   CLLM_PREFILL=f16   D[N, M] = dequantize(W)[N, K] . fp16(X)[M, K]^T, fp32 accumulate:  W (Q4_0 / Q4_1 / Q8_0 / Q4_K) is unpacked and scaled to fp16 while its
//                      tile is staged into LDS -- dequantize_row_*'s values (ggml-quants.c:307-325, 401-414, 1352-1373) rounded to fp16 -- so the dequantized
//                      weights never exist in HBM; X is rounded to fp16 (RNE) by one small pass; v_mfma_f32_32x32x16_f16.
//
// This is NOT the reference CPU computation (which quantizes the activations to Q8_0 / Q8_K and takes integer block dot products): no activation-quantization
// error, an fp16 rounding of the weights instead.  Off by default (the default prefill is the exact-order path, mmx.hip); tests/test_gpu_ops.py checks it
// against the oracle's dequantize + float64 GEMM with a stated tolerance.
//
// Tiling: 128 x 128 or 256 x 256 (weight rows n x tokens m: the launcher's pick, below) per 256-thread workgroup, K walked 64 elements per stage, two LDS stages (one barrier per stage: the next
// stage is written while the current one feeds the matrix cores), global -> registers one stage ahead (raw bytes: nothing converted before the data is
// needed), 144-byte tile rows (conflict-free ds_read_b128 fragments).  A wave owns 2 x 2 or 4 x 4 MFMA tiles (accumulators in AGPRs).  A operand = tokens, B operand = weight rows:
// D[v] = (token (v & 3) + 8 (v >> 2) + 4 (lane >> 5), row lane & 31): stores are 128-byte runs.
#include "common.h"

typedef _Float16 h8v __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct mmd_args {
    const char * W; int64_t nb01; int64_t N; int64_t K;
    const uint16_t * X; int64_t ldx; int64_t M;      // fp16 activations [M][ldx]
    float * dst; int64_t ldd;
    const float * resid; int64_t ldr; int epi;
};
#define MMD_KS 64
#define MMD_LD (MMD_KS * 2 + 16)

__device__ __forceinline__ void b2h(uint32_t u, uint32_t & lo, uint32_t & hi) {       // four unsigned bytes -> fp16 pairs 1024 + byte (0x64uu)
    lo = __builtin_amdgcn_perm(0x64646464u, u, 0x04010400u);
    hi = __builtin_amdgcn_perm(0x64646464u, u, 0x04030402u);
}
__device__ __forceinline__ uint32_t h2op_sub(uint32_t x, uint32_t c) { const h2v r = __builtin_bit_cast(h2v, x) - __builtin_bit_cast(h2v, c); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t h2op_mul(uint32_t x, uint32_t c) { const h2v r = __builtin_bit_cast(h2v, x) * __builtin_bit_cast(h2v, c); return __builtin_bit_cast(uint32_t, r); }
__device__ __forceinline__ uint32_t h2op_fma(uint32_t x, uint32_t a, uint32_t c) {
    const h2v r = __builtin_elementwise_fma(__builtin_bit_cast(h2v, x), __builtin_bit_cast(h2v, a), __builtin_bit_cast(h2v, c));
    return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ uint32_t h2dup(uint16_t h) { return (uint32_t) h * 0x00010001u; }

// BM tokens x BN weight rows per 256-thread workgroup; a wave owns (BM / 2) x (BN / 2) = MI x NJ MFMA tiles.  128 x 128 (2 x 2 tiles per wave, two workgroups per CU):
// one LDS fragment read per MFMA -- the LDS pipe, not the matrix cores, is the busy unit.  256 x 256 (4 x 4 tiles per wave, one workgroup per CU): half a read per MFMA,
// and the dequantization of a weight tile is shared by twice as many tokens.
template <int TYPE, int BM, int BN>
__global__ void __launch_bounds__(256, (BM + BN) <= 256 ? 2 : 1) k_mmd(const mmd_args a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool IS_K = TYPE == CLLM_TYPE_Q4_K, IS_41 = TYPE == CLLM_TYPE_Q4_1, IS_Q8 = TYPE == CLLM_TYPE_Q8_0;
    constexpr int STAGE = (BN + BM) * MMD_LD, WM = BM / 2, WN = BN / 2, MI = WM / 32, NJ = WN / 32, WT = BN / 128, XT = BM / 32;
    constexpr int BS = IS_Q8 ? 34 : IS_41 ? 20 : 18, QOFF = IS_41 ? 4 : 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned mt, nt;
    gemm_tile_of(blockIdx.x, (unsigned)((a.M + BM - 1) / BM), (unsigned)((a.N + BN - 1) / BN), 4, mt, nt);
    const int64_t m0 = (int64_t) mt * BM, n0 = (int64_t) nt * BN;
    const int wn = (wave & 1) * WN, wm = (wave >> 1) * WM;
    const int l31 = lane & 31, l5 = lane >> 5;
    const int64_t K = a.K;

    // the accumulators are NAMED variables, not an array: an array of 16 f32x16 is unrolled too late for the compiler to split it into registers -- it stays a stack
    // object that is written back to scratch memory in every K stage (measured: 3.8x slower)
#define MMD_TILES(F) F(0, 0) F(0, 1) F(0, 2) F(0, 3) F(1, 0) F(1, 1) F(1, 2) F(1, 3) F(2, 0) F(2, 1) F(2, 2) F(2, 3) F(3, 0) F(3, 1) F(3, 2) F(3, 3)
#define MMD_DECL(i, j) [[maybe_unused]] f32x16 c##i##j = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    MMD_TILES(MMD_DECL)
#undef MMD_DECL

    // ---- staging: global -> registers (raw), registers -> LDS (dequantize) ----
    struct __attribute__((packed, aligned(2))) q16 { uint32_t x, y, z, w; };
    const int wrow0 = tid >> 1, whalf = tid & 1;                    // weight tasks: (row wrow0 + 128 w, 32 of the stage's 64 elements)
    u32x4 rq_[WT], rq2_[WT], rh_[WT]; uint32_t rd_[WT];
    u32x4 rx[XT];
    auto prefetch = [&](int64_t k0) {
#pragma unroll
      for (int w = 0; w < WT; w++) {
        u32x4 & rq = rq_[w], & rq2 = rq2_[w], & rh = rh_[w]; uint32_t & rd = rd_[w];
        const int64_t n = n0 + wrow0 + 128 * w, e0 = k0 + 32 * whalf;
        rq = u32x4{0, 0, 0, 0}; rq2 = u32x4{0, 0, 0, 0}; rh = u32x4{0, 0, 0, 0}; rd = 0;
        if (n < a.N && e0 < K) {
            if constexpr (IS_K) {
                const char * bp = a.W + n * a.nb01 + (k0 >> 8) * 144;
                const int c = (int)((k0 & 255) >> 6);
                rh = *(const u32x4 *) bp;
                rq = *(const u32x4 *)(bp + 16 + 32 * c); rq2 = *(const u32x4 *)(bp + 32 + 32 * c);
            } else {
                const char * bp = a.W + n * a.nb01 + (e0 >> 5) * BS;
                if (IS_41) rd = *(const uint32_t *) bp; else rd = *(const uint16_t *) bp;
                const q16 q0 = *(const q16 *)(bp + QOFF);
                rq = u32x4{q0.x, q0.y, q0.z, q0.w};
                if (IS_Q8) { const q16 q1 = *(const q16 *)(bp + 18); rq2 = u32x4{q1.x, q1.y, q1.z, q1.w}; }
            }
        }
      }
#pragma unroll
        for (int t = 0; t < XT; t++) {
            const int c = tid + 256 * t, row = c >> 3, ch = c & 7;
            const int64_t m = m0 + row, e = k0 + ch * 8;
            rx[t] = u32x4{0, 0, 0, 0};
            if (m < a.M && e < K) rx[t] = *(const u32x4 *)(a.X + m * a.ldx + e);      // (K % 8 == 0: whole chunks)
        }
    };
    auto commit = [&](int stage, int64_t k0) {
        char * Wt = lds + stage * STAGE, * Xt = Wt + BN * MMD_LD;
#pragma unroll
      for (int w = 0; w < WT; w++) {
        const u32x4 rq = rq_[w], rq2 = rq2_[w], rh = rh_[w]; const uint32_t rd = rd_[w];
        const int wrow = wrow0 + 128 * w;
        uint32_t o[16];
        if constexpr (IS_K) {
            // sub-block 2 c + whalf of the super-block: y = d * sc * nib - dmin * m (dequantize_row_q4_K), here as one fp16 fma per pair with d * sc and dmin * m rounded to fp16
            const int c = (int)((k0 & 255) >> 6), sbi = 2 * c + whalf;
            const uint32_t u0 = rh.y & 0x3f3f3f3fu, u2 = rh.z & 0x3f3f3f3fu;
            const uint32_t u1 = (rh.w & 0x0f0f0f0fu) | (((rh.y >> 6) & 0x03030303u) << 4);
            const uint32_t u3 = ((rh.w >> 4) & 0x0f0f0f0fu) | (((rh.z >> 6) & 0x03030303u) << 4);
            const float sc = (float)((((sbi & 4) ? u1 : u0) >> (8 * (sbi & 3))) & 0xff), mn = (float)((((sbi & 4) ? u3 : u2) >> (8 * (sbi & 3))) & 0xff);
            const float d = h2f((uint16_t)(rh.x & 0xffff)), dmin = h2f((uint16_t)(rh.x >> 16));
            const uint32_t d1 = h2dup(f2h(d * sc)), m1 = h2dup(f2h(-(dmin * mn))), k1024 = 0x64006400u;
            const uint32_t q[8] = { rq.x, rq.y, rq.z, rq.w, rq2.x, rq2.y, rq2.z, rq2.w };
#pragma unroll
            for (int e = 0; e < 8; e++) {
                uint32_t lo, hi;
                b2h(whalf ? (q[e] >> 4) & 0x0f0f0f0fu : q[e] & 0x0f0f0f0fu, lo, hi);
                o[2 * e] = h2op_fma(h2op_sub(lo, k1024), d1, m1); o[2 * e + 1] = h2op_fma(h2op_sub(hi, k1024), d1, m1);
            }
        } else if constexpr (IS_Q8) {
            const uint32_t dd = h2dup((uint16_t) rd);
            const uint32_t q[8] = { rq.x, rq.y, rq.z, rq.w, rq2.x, rq2.y, rq2.z, rq2.w };
#pragma unroll
            for (int e = 0; e < 8; e++) {
                uint32_t lo, hi;
                b2h(q[e] ^ 0x80808080u, lo, hi);
                o[2 * e] = h2op_mul(h2op_sub(lo, 0x64806480u), dd); o[2 * e + 1] = h2op_mul(h2op_sub(hi, 0x64806480u), dd);
            }
        } else {
            const uint32_t dd = h2dup((uint16_t)(rd & 0xffff)), mm = h2dup((uint16_t)(rd >> 16));
            const uint32_t q[4] = { rq.x, rq.y, rq.z, rq.w };
#pragma unroll
            for (int e = 0; e < 4; e++) {                               // elements 4 e .. (low nibbles), 16 + 4 e .. (high nibbles)
                uint32_t lo, hi, lo2, hi2;
                b2h(q[e] & 0x0f0f0f0fu, lo, hi); b2h((q[e] >> 4) & 0x0f0f0f0fu, lo2, hi2);
                if (IS_41) {
                    o[2 * e] = h2op_fma(h2op_sub(lo, 0x64006400u), dd, mm); o[2 * e + 1] = h2op_fma(h2op_sub(hi, 0x64006400u), dd, mm);
                    o[8 + 2 * e] = h2op_fma(h2op_sub(lo2, 0x64006400u), dd, mm); o[8 + 2 * e + 1] = h2op_fma(h2op_sub(hi2, 0x64006400u), dd, mm);
                } else {
                    o[2 * e] = h2op_mul(h2op_sub(lo, 0x64086408u), dd); o[2 * e + 1] = h2op_mul(h2op_sub(hi, 0x64086408u), dd);
                    o[8 + 2 * e] = h2op_mul(h2op_sub(lo2, 0x64086408u), dd); o[8 + 2 * e + 1] = h2op_mul(h2op_sub(hi2, 0x64086408u), dd);
                }
            }
        }
        char * wr = Wt + wrow * MMD_LD + whalf * 64;
#pragma unroll
        for (int e = 0; e < 4; e++) *(u32x4 *)(wr + 16 * e) = u32x4{o[4 * e], o[4 * e + 1], o[4 * e + 2], o[4 * e + 3]};
      }
#pragma unroll
        for (int t = 0; t < XT; t++) {
            const int c = tid + 256 * t, row = c >> 3, ch = c & 7;
            *(u32x4 *)(Xt + row * MMD_LD + ch * 16) = rx[t];
        }
    };

    // (measured dead end: the next stage's dequantization placed between the MFMAs of the current one with a sched_group_barrier pipeline -- 128 x 128: 1481 -> 1766 us
    //  on the gate/up GEMM, 256 x 256: hundreds of spilled registers)
    prefetch(0);
    commit(0, 0);
    __syncthreads();
    int stage = 0;
    for (int64_t k0 = 0; k0 < K; k0 += MMD_KS) {
        const bool more = k0 + MMD_KS < K;
        if (more) prefetch(k0 + MMD_KS);
        const char * Wt = lds + stage * STAGE, * Xt = Wt + BN * MMD_LD;
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            h8v ax[MI], bw[NJ];
#pragma unroll
            for (int i = 0; i < MI; i++) ax[i] = *(const h8v *)(Xt + (wm + i * 32 + l31) * MMD_LD + ks * 32 + l5 * 16);
#pragma unroll
            for (int j = 0; j < NJ; j++) bw[j] = *(const h8v *)(Wt + (wn + j * 32 + l31) * MMD_LD + ks * 32 + l5 * 16);
#define MMD_MFMA(i, j) if constexpr (i < MI && j < NJ) c##i##j = __builtin_amdgcn_mfma_f32_32x32x16_f16(ax[i < MI ? i : 0], bw[j < NJ ? j : 0], c##i##j, 0, 0, 0);
            MMD_TILES(MMD_MFMA)
#undef MMD_MFMA
        }
        if (more) commit(stage ^ 1, k0 + MMD_KS);
        __syncthreads();
        stage ^= 1;
    }

    const int64_t nv = (a.N / 2) & ~(int64_t) 7;
    auto store_tile = [&](int i, int j, const f32x16 & c) {
        const int64_t n = n0 + wn + j * 32 + l31;
        if (a.epi == 1) {
            const int64_t u = n >> 1;
            const bool nok = !(lane & 1) && n + 1 < a.N, body = u < nv;
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int64_t m = m0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * l5;
                const float r = c[v];
                const float up = dpp_f<DPP_QUAD_XOR1>(r);
                if (nok && m < a.M) a.dst[m * a.ldd + u] = silu_any(r, body) * up;
            }
        } else {
#pragma unroll
            for (int v = 0; v < 16; v++) {
                const int64_t m = m0 + wm + i * 32 + (v & 3) + 8 * (v >> 2) + 4 * l5;
                const float r = c[v];
                if (n < a.N && m < a.M) a.dst[m * a.ldd + n] = a.resid ? r + a.resid[m * a.ldr + n] : r;
            }
        }
    };
#define MMD_STORE(i, j) if constexpr (i < MI && j < NJ) store_tile(i, j, c##i##j);
    MMD_TILES(MMD_STORE)
#undef MMD_STORE
#undef MMD_TILES
}

__global__ void __launch_bounds__(256) k_f32_to_f16(const char * __restrict__ x, int64_t nb1, int64_t K, uint16_t * __restrict__ out, int64_t ldo) {
    const int64_t e = ((int64_t) blockIdx.x * 256 + threadIdx.x) * 2;
    const int64_t row = blockIdx.y;
    if (e >= K) return;
    const float * r = (const float *)(x + row * nb1);
    *(uint32_t *)(out + row * ldo + e) = (uint32_t) f2h(r[e]) | ((uint32_t) f2h(e + 1 < K ? r[e + 1] : 0.0f) << 16);
}

static int g_f16_mode = -1;
bool prefill_f16_enabled() { if (g_f16_mode < 0) g_f16_mode = getenv("CLLM_PREFILL") && !strcmp(getenv("CLLM_PREFILL"), "f16"); return g_f16_mode != 0; }
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_prefill_f16(int on) { g_f16_mode = on ? 1 : 0; }      // tests: switch inside one process
static int g_mmd_tile = -1;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_mmd_tile(int tile) { g_mmd_tile = tile; }               // tests / tools: 0 pick, 128, 256

// w: [K, N] quantized rows (2-D), x: [K, M] f32 rows (nb1 stride), d: [N, M] f32 (nb1 stride); CLLM_E_UNSUPPORTED -> the caller's other paths take it
int launch_dense_f16(hipStream_t st, int wtype, const tview & w, const tview & x, const tview & d, const float * resid, int64_t ldr, int epi) {
    const int64_t K = w.ne[0], N = w.ne[1], M = x.ne[1];
    if (K % 32 || d.nb[1] % 4 || x.nb[0] != 4 || (epi && (epi != 1 || resid || N % 2)) || (wtype == CLLM_TYPE_Q4_K && ((uintptr_t) w.data % 16 || w.nb[1] % 16))) return CLLM_E_UNSUPPORTED;
    if (wtype == CLLM_TYPE_Q4_1 && ((uintptr_t) w.data % 4 || w.nb[1] % 4)) return CLLM_E_UNSUPPORTED;
    if (((M + 127) / 128) * ((N + 127) / 128) > 0x7fffffff) return CLLM_E_UNSUPPORTED;
    const int64_t ldx = (K + 7) & ~(int64_t) 7;
    const size_t need = (size_t) M * ldx * 2;
    void * g_x16 = stream_scratch(st, SCRATCH_X16, need);      // the fp16 activation copy: per (device, stream), outgrown blocks kept for captured launch lists
    if (!g_x16) return CLLM_E_HIP;
    hipLaunchKernelGGL(k_f32_to_f16, dim3((unsigned)((K / 2 + 255) / 256), (unsigned) M), dim3(256), 0, st, (const char *) x.data, x.nb[1], K, (uint16_t *) g_x16, ldx);
    LAUNCH_CHECK();
    mmd_args a;
    a.W = w.data; a.nb01 = w.nb[1]; a.N = N; a.K = K; a.X = (const uint16_t *) g_x16; a.ldx = ldx; a.M = M;
    a.dst = (float *) d.data; a.ldd = d.nb[1] / 4; a.resid = resid; a.ldr = ldr; a.epi = epi;
    // the tile: 128 x 128.  256 x 256 (CLLM_MMD_TILE=256 / cllm_debug_set_mmd_tile) is faster per GEMM in isolation where its workgroups fill the CUs evenly
    // (o 254 -> 236 us, down 861 -> 786 us at 4096 tokens) but not in the running prefill: cfg3 113.8 ms with 128 x 128, 114.0 picked per shape, 117.5 ms with
    // 256 x 256 everywhere (tools/prefill_f16_tile_ab.py, one process) -- so it is not picked.
    if (g_mmd_tile < 0) g_mmd_tile = getenv("CLLM_MMD_TILE") ? atoi(getenv("CLLM_MMD_TILE")) : 0;  // tests / tools: 128 / 256 force
    const int64_t big_tiles = ((M + 255) / 256) * ((N + 255) / 256);
    const bool big = g_mmd_tile == 256;
#define GO(T) do { \
        if (big) { constexpr int LDS = 2 * 512 * MMD_LD; static uint64_t attr = 0; \
            if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_mmd<T, 256, 256>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); dev_flag_set(attr); } \
            hipLaunchKernelGGL((k_mmd<T, 256, 256>), dim3((unsigned) big_tiles), dim3(256), LDS, st, a); } \
        else { constexpr int LDS = 2 * 256 * MMD_LD; static uint64_t attr = 0; \
            if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_mmd<T, 128, 128>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS)); dev_flag_set(attr); } \
            hipLaunchKernelGGL((k_mmd<T, 128, 128>), dim3((unsigned)(((M + 127) / 128) * ((N + 127) / 128))), dim3(256), LDS, st, a); } } while (0)
    if (wtype == CLLM_TYPE_Q4_K) GO(CLLM_TYPE_Q4_K);
    else if (wtype == CLLM_TYPE_Q4_0) GO(CLLM_TYPE_Q4_0);
    else if (wtype == CLLM_TYPE_Q8_0) GO(CLLM_TYPE_Q8_0);
    else if (wtype == CLLM_TYPE_Q4_1) GO(CLLM_TYPE_Q4_1);
    else return CLLM_E_UNSUPPORTED;
#undef GO
    LAUNCH_CHECK();
    return CLLM_OK;
}
