// gemv_rows32.hip -- the decode step's mat-vec for the 32-weight block formats (Q4_0 / Q4_1 / Q8_0): a wave streams RPW rows side by side.
//
// Same contract as k_gemv_dec (gemv_decode_kernel.h): activation produced in the kernel (prologue 1..4), dst = W . act (+ bias) (+ resid) or the SiLU(gate) * up
// epilogue, every row accumulated in the ORDER of the reference's AVX2 ggml_vec_dot_q4_0_q8_0 / q4_1_q8_1 / q8_0_q8_0 (arch/x86/quants.c:543-577, 701-760, 1012-1040):
// per block b in row order and AVX lane A, acc[A] = fma(d_w d_x, (float) sumi[A], acc[A]) -- bit-identical to libggml-cpu.so.
//
// Why a second form: these formats fold the scale in every 32 weights (Q4_K: every 256), i.e. EIGHT serial fp32 chain steps per 18 / 20 / 34 bytes.  k_gemv_dec gives a
// row to a whole wave: one lane per block makes the eight integer sums and writes chain records to LDS, then lanes 0..7 walk the 64 blocks of records serially: 64 fma +
// 32 LDS reads with 8 of 64 lanes at work -- more instructions than the integer part itself, and the launch is instruction-issue bound (0.28 of the HBM roofline on
// Q4_0 gate/up).  Here a step is RPW rows x LPR = 64 / RPW blocks: lane (r, t) owns block t of row r in a step (same loads, same integer work), the records are
// [row][slot of A][t], and EVERY lane runs a chain: lane (r, j) walks the LPR blocks of row r for slot j -- LPR fma per step instead of 64 (RPW 8: 8).
// RPW is picked by the launcher so that there are at least ~8 units per CU (gate/up, lm_head: 8; the hidden-sized outputs: 2 or 4).
#include "common.h"
#include "quant_dev.h"
#include "q4k.h"
#include "q32.h"

#ifndef R32_P
#define R32_P 4
#endif

typedef int r32_i4 __attribute__((ext_vector_type(4)));
#define R32_REC_BYTES (64 * 9 * 4 + 2 * 256)          // per wave: RPW rows x 9 sub-rows (8 slots + d_w d_x) x LPR floats, then Q4_1's m_w[64], s_a[64]

__device__ __forceinline__ float r32_silu(float x) { return x / (1.0f + ggml_expf_poly(0.0f - x)); }
__device__ __forceinline__ float r32_silu_any(float x, bool body) { return body ? r32_silu(x) : x / (1.0f + libm_expf(-x)); }

template <int FMT, int PRO, int EPI, int NPRE, int RPW>
__global__ void __launch_bounds__(1024) k_gemv_rows32(const float * __restrict__ px, const float * __restrict__ pw, const char * __restrict__ W, int nblk, int nunits, float eps,
                                                      float * __restrict__ dst, const float * __restrict__ bias, const float * resid) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr bool IS_Q8 = FMT == CLLM_TYPE_Q8_0, IS_41 = FMT == CLLM_TYPE_Q4_1;
    constexpr int BS = q32_fmt<FMT>::BS, P = R32_P, LPR = 64 / RPW;
    const int tid = threadIdx.x, lane = tid & 63;
    const int K = nblk * 32;
    const unsigned nb01 = (unsigned) nblk * (unsigned) BS;
#ifndef R32_C0_PLANE
#define R32_C0_PLANE 0                                 // 1: Q4_0's (-8, -8, -8, -8) . a per (block, AVX lane) from a plane the prologue leaves in LDS; 0: taken on the fly (the LDS pipe is the busy unit)
#endif
    constexpr bool IS_40 = !IS_Q8 && !IS_41, NEED_C0 = IS_40 && R32_C0_PLANE;
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

    // ---- (2) this wave's units (RPW consecutive rows): (k * 16 + wave) * grid + block; the first P steps fly during the prologue.
    //          A block's bytes are loaded as the 4-byte aligned window around it (q32.h); rows are whole dwords, so a block starts in the upper half of its
    //          first dword exactly when its index is odd: the lane's byte selectors are constants ----
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int S = (nblk + LPR - 1) / LPR;                              // steps per unit
    const int ustride = 16 * gridDim.x, u0 = wave * gridDim.x + blockIdx.x;
    const int nmine = u0 < nunits ? (nunits - u0 + ustride - 1) / ustride : 0;
    const int r = lane / LPR, t = lane % LPR;
    const bool odd = !IS_41 && (t & 1);
    const uint32_t psel = odd ? 0x07060504u : 0x05040302u;             // v_perm_b32 selector: bytes 2..5 / 4..7 of a dword pair
    const unsigned lane_off = (unsigned) r * nb01 + (unsigned) t * BS - (odd ? 2u : 0u);
    u32x4 qa[P], qb[IS_Q8 ? P : 1];
    uint32_t qt[P];
    int iu = 0, is = 0;
    auto issue = [&](int p) {                                          // unconditional: out-of-range lanes re-read the unit's first block and are masked
        const bool ok = iu < nmine && LPR * is + t < nblk;
#ifdef R32_FAKE_COALESCED        /* timing experiment only (wrong results): every lane 16 contiguous, aligned bytes of the step's region */
        const char * bp = W + (size_t)(unsigned)(ok ? u0 + iu * ustride : u0 < nunits ? u0 : 0) * (size_t)(RPW * nb01) + (ok ? (unsigned)(is * 1152 + lane * 16) : 0u);
#else
        const char * bp = W + (size_t)(unsigned)(ok ? u0 + iu * ustride : u0 < nunits ? u0 : 0) * (size_t)(RPW * nb01) + (ok ? lane_off + (unsigned)(LPR * is) * BS : 0u);
#endif
        if (IS_41) { qt[p] = *(const uint32_t *) bp; qa[p] = *(const u32x4 *)(bp + 4); }
        else {
            qa[p] = *(const u32x4 *) bp;
            if (IS_Q8) { qb[IS_Q8 ? p : 0] = *(const u32x4 *)(bp + 16); qt[p] = *(const uint32_t *)(bp + 32); }
            else qt[p] = *(const uint32_t *)(bp + 16);
        }
        if (++is == S) { is = 0; iu++; }
    };
#pragma unroll
    for (int p = 0; p < P; p++) issue(p);

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
    if (nmine == 0) return;

    // ---- (4) stream the units ----
    const int j = lane & 7;                                            // the chain this lane runs: slot j of row r
    const char * act = lds;
    const float * actd = (const float *)(lds + act_off_d(K));
    const float * acts = (const float *)(lds + act_off_s(K, IS_41 ? ACT_Q8_1 : ACT_Q8_0));
    float * rec = (float *)(lds + arb + (NEED_C0 ? K : 0) + wave * R32_REC_BYTES);
    float * X = rec + r * (9 * LPR) + t;                               // this lane's record column: X[slot * LPR], X[8 * LPR] = d_w d_x
    float * M = rec + 64 * 9;                                          // Q4_1: m_w[lane], s_a[64 + lane]
    const float * xr = rec + r * (9 * LPR) + j * LPR, * dr = rec + r * (9 * LPR) + 8 * LPR, * mr = M + r * LPR;
    float acc = 0.0f, accs = 0.0f;
    int cu = 0, cs = 0;
    while (cu < nmine) {
#pragma unroll
        for (int p = 0; p < P; p++) {
            const int b = LPR * cs + t;
            const bool ok = cu < nmine && b < nblk;
            const int bb = ok ? b : 0;
            // -- the block in place: h = fp16 d (Q4_1: | m << 16), q0 (q1) = the quant bytes
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
            issue(p);
            // -- eight exact integer sums (q32_emit's arithmetic), records
            const u32x4 a0 = *(const u32x4 *)(act + bb * 32), a1 = *(const u32x4 *)(act + bb * 32 + 16);
            const float yd = actd[bb];
            int s[8];
            if (IS_Q8) {
                s[0] = dot4(q0.x, a0.x, 0); s[1] = dot4(q0.y, a0.y, 0); s[2] = dot4(q0.z, a0.z, 0); s[3] = dot4(q0.w, a0.w, 0);
                s[4] = dot4(q1.x, a1.x, 0); s[5] = dot4(q1.y, a1.y, 0); s[6] = dot4(q1.z, a1.z, 0); s[7] = dot4(q1.w, a1.w, 0);
            } else {
                r32_i4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};           // Q4_0: the prologue's c0 plane
                if (NEED_C0) { c0 = *(const r32_i4 *)(act + arb + bb * 32); c1 = *(const r32_i4 *)(act + arb + bb * 32 + 16); }
                else if (IS_40) {
                    c0 = r32_i4{dot4(0xf8f8f8f8u, a0.x, 0), dot4(0xf8f8f8f8u, a0.y, 0), dot4(0xf8f8f8f8u, a0.z, 0), dot4(0xf8f8f8f8u, a0.w, 0)};
                    c1 = r32_i4{dot4(0xf8f8f8f8u, a1.x, 0), dot4(0xf8f8f8f8u, a1.y, 0), dot4(0xf8f8f8f8u, a1.z, 0), dot4(0xf8f8f8f8u, a1.w, 0)};
                }
                s[0] = dot4(q0.x & 0x0f0f0f0fu, a0.x, c0.x); s[1] = dot4(q0.y & 0x0f0f0f0fu, a0.y, c0.y);
                s[2] = dot4(q0.z & 0x0f0f0f0fu, a0.z, c0.z); s[3] = dot4(q0.w & 0x0f0f0f0fu, a0.w, c0.w);
                s[4] = dot4((q0.x >> 4) & 0x0f0f0f0fu, a1.x, c1.x); s[5] = dot4((q0.y >> 4) & 0x0f0f0f0fu, a1.y, c1.y);
                s[6] = dot4((q0.z >> 4) & 0x0f0f0f0fu, a1.z, c1.z); s[7] = dot4((q0.w >> 4) & 0x0f0f0f0fu, a1.w, c1.w);
            }
            // slot of AVX lane A = 4(A&1) + (A&2) + (A>>2): [A0 A4 A2 A6 | A1 A5 A3 A7] (the hsum below)
#ifdef R32_FAKE_NOREC
            acc += __int_as_float((s[0] ^ s[1] ^ s[2] ^ s[3] ^ s[4] ^ s[5] ^ s[6] ^ s[7]) & 1) * yd + (ok ? h2f((uint16_t) h) : 0.0f);
#else
            X[0 * LPR] = (float) s[0]; X[1 * LPR] = (float) s[4]; X[2 * LPR] = (float) s[2]; X[3 * LPR] = (float) s[6];
            X[4 * LPR] = (float) s[1]; X[5 * LPR] = (float) s[5]; X[6 * LPR] = (float) s[3]; X[7 * LPR] = (float) s[7];
            X[8 * LPR] = ok ? h2f((uint16_t) h) * yd : 0.0f;          // a masked block: fma(0, finite, acc) = acc
            if (IS_41) { M[lane] = ok ? h2f((uint16_t)(h >> 16)) : 0.0f; M[64 + lane] = acts[bb]; }
#endif
            wave_lds_fence();
            // -- the chains: LPR blocks in row order
#ifndef R32_FAKE_NOCHAIN       /* timing experiments only (wrong results): R32_FAKE_NOCHAIN, R32_FAKE_NOREC, R32_FAKE_COALESCED */
#pragma unroll
            for (int i = 0; i < LPR; i += 4) {
                const f32x4 xv = *(const f32x4 *)(xr + i), dv = *(const f32x4 *)(dr + i);
                acc = __builtin_fmaf(dv.x, xv.x, acc); acc = __builtin_fmaf(dv.y, xv.y, acc); acc = __builtin_fmaf(dv.z, xv.z, acc); acc = __builtin_fmaf(dv.w, xv.w, acc);
                if (IS_41) {                                          // summs = fma(m_w, s_a, summs) (arch/x86/quants.c:726): the same chain in every lane of the row
                    const f32x4 mv = *(const f32x4 *)(mr + i), sv = *(const f32x4 *)(mr + 64 + i);
                    accs = __builtin_fmaf(mv.x, sv.x, accs); accs = __builtin_fmaf(mv.y, sv.y, accs); accs = __builtin_fmaf(mv.z, sv.z, accs); accs = __builtin_fmaf(mv.w, sv.w, accs);
                }
            }
#endif
            wave_lds_fence();
            if (++cs == S) {                                          // RPW rows complete: hsum_float_8 over the 8 slots (neighbour exchanges), epilogue, store
                float hsum = acc;
                hsum = hsum + dpp_f<DPP_QUAD_XOR1>(hsum); hsum = hsum + dpp_f<DPP_QUAD_XOR2>(hsum); hsum = hsum + dpp_f<DPP_HALF_MIRROR>(hsum);
                float v = IS_41 ? hsum + accs : hsum;
                if (cu < nmine) {
                    const int unit = u0 + cu * ustride;
                    if (EPI == 1) {                                   // rows alternate gate_u, up_u: the up row is the next lane group
                        const float up = __int_as_float(__builtin_amdgcn_ds_bpermute(((lane + LPR) & 63) * 4, __float_as_int(v)));
                        if (t == 0 && !(r & 1)) dst[unit * (RPW / 2) + (r >> 1)] = r32_silu(v) * up;
                    } else {
                        if (bias || resid) {                          // the unit's RPW values through the scalar cache (no wait on the weight prefetch)
                            float bsel = 0.0f, rsel = 0.0f;
#pragma unroll
                            for (int q = 0; q < RPW; q++) {
                                if (bias)  { const float x = uniform_load_f32(bias  + (size_t) unit * RPW + q); bsel = r == q ? x : bsel; }
                                if (resid) { const float x = uniform_load_f32(resid + (size_t) unit * RPW + q); rsel = r == q ? x : rsel; }
                            }
                            if (bias)  v = v + bsel;
                            if (resid) v = v + rsel;
                        }
                        if (t == 0) dst[(size_t) unit * RPW + r] = v;
                    }
                }
                acc = 0.0f; accs = 0.0f; cs = 0; cu++;
            }
        }
    }
}

static int g_rows32_mode = -1;
extern "C" __attribute__((visibility("default"))) void cllm_debug_set_gemv_rows32(int mode) { g_rows32_mode = mode; }      // tests / tools: 0 = k_gemv_dec, 1 = pick, 2 / 4 / 8 = force

// K % 32 == 0, rows whole dwords (Q4_0 / Q8_0: K % 64 == 0), nrows % RPW == 0; CLLM_E_UNSUPPORTED: k_gemv_dec takes the launch
int launch_gemv_rows32(hipStream_t st, int wtype, const void * W, int64_t K, int64_t nrows, int pro, const float * px, const float * pw, float eps, int epi, float * dst,
                       const float * bias, const float * resid) {
    if (g_rows32_mode < 0) g_rows32_mode = getenv("CLLM_GEMV_ROWS32") ? atoi(getenv("CLLM_GEMV_ROWS32")) : 1;      // 0: off, 2 / 4 / 8: that many rows per wave where the shape allows
    const int mode = g_rows32_mode;
    if (!mode || (wtype != CLLM_TYPE_Q4_0 && wtype != CLLM_TYPE_Q4_1 && wtype != CLLM_TYPE_Q8_0)) return CLLM_E_UNSUPPORTED;
    const int bs = wtype == CLLM_TYPE_Q8_0 ? 34 : wtype == CLLM_TYPE_Q4_1 ? 20 : 18;
    if (K % 32 || ((K / 32) * bs) % 4 || pro < 1 || pro > 4 || nrows <= 0 || ((uintptr_t) W & 3) || (uint64_t) nrows * (uint64_t)(K / 32 * bs) >= (1ull << 32)) return CLLM_E_UNSUPPORTED;
    if (K > ((pro == 2 || pro == 4) ? 32768 : 16384)) return CLLM_E_UNSUPPORTED;
    if (epi == 1 && (pro != 1 || bias || resid || nrows % 2)) return CLLM_E_UNSUPPORTED;
    const int nblk = (int)(K / 32), cus = device_cu_count();
    // rows per wave: 8 where that leaves >= 8 units per CU (gate/up, lm_head: 17.5 vs 29 us, 63 vs 120 us on Llama-3-8B Q4_0); the hidden-sized outputs have too few
    // rows for that -- one row per wave (k_gemv_dec) keeps 16 waves per CU busy and measured faster than 2 or 4 rows side by side (o: 7.3 vs 10.4 us, down: 21 vs 33 us).
    // 2 / 4 are kept for the tests and tools (cllm_debug_set_gemv_rows32 / CLLM_GEMV_ROWS32).
    int rpw = 0;
    if (mode == 1) { if (pro == 1 && nrows % 8 == 0 && nrows / 8 >= 8 * (int64_t) cus) rpw = 8; }
    else if ((mode == 2 || mode == 4 || mode == 8) && nrows % mode == 0 && (mode != 8 || pro == 1)) rpw = mode;
    if (!rpw) return CLLM_E_UNSUPPORTED;
    const int nunits = (int)(nrows / rpw);
    const int grid = (nunits + 15) / 16 < cus ? (nunits + 15) / 16 : cus;
    const size_t lds = act_row_bytes(K, wtype == CLLM_TYPE_Q4_1 ? ACT_Q8_1 : ACT_Q8_0) + ((wtype == CLLM_TYPE_Q4_0 && R32_C0_PLANE) ? (size_t) K : 0) + 16 * (size_t) R32_REC_BYTES;
    if (lds > 159 * 1024) return CLLM_E_UNSUPPORTED;
    const int npre = K <= 4096 ? 1 : K <= 16384 ? 4 : 8;
#define GOR(FMT_, PRO_, EPI_, NPRE_, RPW_) do { \
        static uint64_t attr = 0; \
        if (dev_flag_unset(attr)) { HIP_TRY(hipFuncSetAttribute((const void *) k_gemv_rows32<FMT_, PRO_, EPI_, NPRE_, RPW_>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024)); dev_flag_set(attr); } \
        hipLaunchKernelGGL((k_gemv_rows32<FMT_, PRO_, EPI_, NPRE_, RPW_>), dim3((unsigned) grid), dim3(1024), lds, st, px, pw, (const char *) W, nblk, nunits, eps, dst, bias, resid); } while (0)
#define GOW(FMT_, PRO_, EPI_, NPRE_) do { if (rpw == 2) GOR(FMT_, PRO_, EPI_, NPRE_, 2); else GOR(FMT_, PRO_, EPI_, NPRE_, 4); } while (0)
#define GOW1(FMT_, EPI_, NPRE_) do { if (rpw == 8) GOR(FMT_, 1, EPI_, NPRE_, 8); else GOW(FMT_, 1, EPI_, NPRE_); } while (0)
#define GOP(FMT_) do { \
        if (pro == 1 && epi == 1) { if (npre == 1) GOW1(FMT_, 1, 1); else GOW1(FMT_, 1, 4); } \
        else if (pro == 1)        { if (npre == 1) GOW1(FMT_, 0, 1); else GOW1(FMT_, 0, 4); } \
        else if (pro == 2)        { if (npre == 1) GOW(FMT_, 2, 0, 1); else if (npre == 4) GOW(FMT_, 2, 0, 4); else GOW(FMT_, 2, 0, 8); } \
        else if (pro == 4)        { if (npre == 1) GOW(FMT_, 4, 0, 1); else if (npre == 4) GOW(FMT_, 4, 0, 4); else GOW(FMT_, 4, 0, 8); } \
        else                      { if (npre == 1) GOW(FMT_, 3, 0, 1); else GOW(FMT_, 3, 0, 4); } } while (0)
    if (wtype == CLLM_TYPE_Q4_0) GOP(CLLM_TYPE_Q4_0); else if (wtype == CLLM_TYPE_Q4_1) GOP(CLLM_TYPE_Q4_1); else GOP(CLLM_TYPE_Q8_0);
#undef GOP
#undef GOW1
#undef GOW
#undef GOR
    LAUNCH_CHECK();
    return CLLM_OK;
}
