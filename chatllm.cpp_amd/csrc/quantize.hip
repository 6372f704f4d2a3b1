// quantize.hip -- on-device activation quantizers (F32 -> Q8_0 / Q8_K "act rows", layout in common.h).
//
// Reproduces exactly what the reference CPU path does to src1 before its integer dot products:
//   Q8_0: ggml/src/ggml-cpu/arch/x86/quants.c:290-345  d = amax/127 (stored as fp16), id = 127/amax,
//         q = round-half-even(x*id)
//   Q8_K: ggml/src/ggml-quants.c:2555-2592             iscale = -127/max (max = FIRST element of largest |x|, signed),
//         q = min(127, nearest_int(iscale*x)), d = 1/iscale, bsums
// All results are bit-exact w.r.t. the CPU (checked in tests/test_quantize.py).
#include "common.h"

// 8 lanes per 32-block, 4 elements per lane, 8 blocks per wave.
__global__ void __launch_bounds__(256) k_quantize_q8_0(const char * __restrict__ src, int64_t K, int64_t ne1, int64_t ne2,
                                                       int64_t nb1, int64_t nb2, int64_t nb3,
                                                       char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.y;                        // flattened i11 + ne11*(i12 + ne12*i13)
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *)(src + i1*nb1 + i2*nb2 + i3*nb3);
    char * a = act + row * act_stride;
    int8_t *  qs = (int8_t *) a;
    float *   dd = (float *)(a + act_off_d(K));
    int32_t * ss = (int32_t *)(a + act_off_s(K, 32));

    const int64_t e0 = ((int64_t) blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (e0 >= K) return;                                   // K % 32 == 0 -> whole 8-lane groups drop out together
    const f32x4 v = *(const f32x4 *)(x + e0);
    float amax = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    amax = fmaxf(amax, __shfl_xor(amax, 1, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 2, 64));
    amax = fmaxf(amax, __shfl_xor(amax, 4, 64));
    const float d  = amax / 127.f;
    const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
    const int q0 = (int) rintf(v.x * id), q1 = (int) rintf(v.y * id), q2 = (int) rintf(v.z * id), q3 = (int) rintf(v.w * id);
    *(uint32_t *)(qs + e0) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    int s = q0 + q1 + q2 + q3;
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if ((threadIdx.x & 7) == 0) {
        dd[e0 / 32] = h2f(f2h(d));                         // the CPU stores d as fp16 and reads it back for the dot
        ss[e0 / 32] = s;
    }
}

__device__ __forceinline__ int nearest_int(float fval) {   // ggml-quants.c:436-441
    const float val = fval + 12582912.f;
    return (int)(__float_as_uint(val) & 0x007fffff) - 0x00400000;
}

// one wave per 256-block, 4 elements per lane.
__global__ void __launch_bounds__(256) k_quantize_q8_K(const char * __restrict__ src, int64_t K, int64_t ne1, int64_t ne2,
                                                       int64_t nb1, int64_t nb2, int64_t nb3,
                                                       char * __restrict__ act, size_t act_stride) {
    const int64_t row = blockIdx.y;
    const int64_t i1 = row % ne1, i2 = (row / ne1) % ne2, i3 = row / (ne1 * ne2);
    const float * x = (const float *)(src + i1*nb1 + i2*nb2 + i3*nb3);
    char * a = act + row * act_stride;
    int8_t *  qs = (int8_t *) a;
    float *   dd = (float *)(a + act_off_d(K));
    int32_t * ss = (int32_t *)(a + act_off_s(K, 256));

    const int lane = threadIdx.x & 63;
    const int64_t blk = (int64_t) blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
    if (blk * 256 >= K) return;
    const int64_t e0 = blk * 256 + lane * 4;
    const f32x4 v = *(const f32x4 *)(x + e0);

    // first element of largest magnitude: key = (|x| bits, ~index) maximised
    unsigned long long key = 0;
    {
        const float ax[4] = { fabsf(v.x), fabsf(v.y), fabsf(v.z), fabsf(v.w) };
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const unsigned long long k2 = ((unsigned long long) __float_as_uint(ax[i]) << 32) | (unsigned long long)(0xffffffffu - (uint32_t)(lane * 4 + i));
            key = k2 > key ? k2 : key;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(key, o, 64);
        key = other > key ? other : key;
    }
    const float amax = __uint_as_float((uint32_t)(key >> 32));
    const int   imax = (int)(0xffffffffu - (uint32_t) key);
    // fetch the signed value of that element from its owner lane
    const float mine = (imax & 3) == 0 ? v.x : (imax & 3) == 1 ? v.y : (imax & 3) == 2 ? v.z : v.w;
    const float maxv = __shfl(mine, imax >> 2, 64);

    int q0 = 0, q1 = 0, q2 = 0, q3 = 0;
    float d = 0.0f;
    if (amax != 0.0f) {
        const float iscale = -127.f / maxv;
        q0 = min(127, nearest_int(iscale * v.x));
        q1 = min(127, nearest_int(iscale * v.y));
        q2 = min(127, nearest_int(iscale * v.z));
        q3 = min(127, nearest_int(iscale * v.w));
        d = 1 / iscale;
    }
    *(uint32_t *)(qs + e0) = (uint32_t)(q0 & 0xff) | ((uint32_t)(q1 & 0xff) << 8) | ((uint32_t)(q2 & 0xff) << 16) | ((uint32_t)(q3 & 0xff) << 24);
    int s = q0 + q1 + q2 + q3;
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    s += __shfl_xor(s, 4, 64);
    if ((lane & 7) == 0) ss[e0 / 32] = s;
    if (lane == 0) dd[blk] = d;
}

int launch_quantize_act(hipStream_t st, int kind_blk, const tview & s, void * act, size_t act_stride) {
    const int64_t K = s.ne[0];
    const int64_t rows = s.ne[1] * s.ne[2] * s.ne[3];
    if (K % kind_blk) FAIL(CLLM_E_INVALID, "quantize_act: K=%lld not a multiple of %d", (long long) K, kind_blk);
    if (rows <= 0 || K <= 0) return CLLM_OK;
    if (rows > 65535) FAIL(CLLM_E_UNSUPPORTED, "quantize_act: too many rows (%lld)", (long long) rows);
    if (kind_blk == 32) {
        dim3 grid((unsigned)((K / 4 + 255) / 256), (unsigned) rows);
        hipLaunchKernelGGL(k_quantize_q8_0, grid, dim3(256), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], (char *) act, act_stride);
    } else {
        dim3 grid((unsigned)((K / 256 + 3) / 4), (unsigned) rows);
        hipLaunchKernelGGL(k_quantize_q8_K, grid, dim3(256), 0, st, s.data, K, s.ne[1], s.ne[2], s.nb[1], s.nb[2], s.nb[3], (char *) act, act_stride);
    }
    LAUNCH_CHECK();
    return CLLM_OK;
}

// ---- KAT surface: act row -> reference block layout ------------------------------------------------
__global__ void k_act_to_q8_0_blocks(const char * __restrict__ act, int64_t K, block_q8_0 * __restrict__ y) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K / 32) return;
    const int8_t * qs = (const int8_t *) act;
    const float *  dd = (const float *)(act + act_off_d(K));
    y[b].d = f2h(dd[b]);
    for (int j = 0; j < 32; j++) y[b].qs[j] = qs[b*32 + j];
}
__global__ void k_act_to_q8_K_blocks(const char * __restrict__ act, int64_t K, block_q8_K * __restrict__ y) {
    const int64_t b = (int64_t) blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= K / 256) return;
    const int8_t * qs = (const int8_t *) act;
    const float *  dd = (const float *)(act + act_off_d(K));
    y[b].d = dd[b];
    for (int j = 0; j < 16; j++) {
        int s = 0;
        for (int i = 0; i < 16; i++) { const int8_t q = qs[b*256 + j*16 + i]; y[b].qs[j*16 + i] = q; s += q; }
        y[b].bsums[j] = (int16_t) s;
    }
}

static int quantize_to_blocks(void * stream, int kind_blk, const float * x, void * y, int64_t k) {
    hipStream_t st = (hipStream_t) stream;
    if (!x || !y || k <= 0 || k % kind_blk) FAIL(CLLM_E_INVALID, "quantize_row: bad arguments");
    void * act = nullptr;
    const size_t bytes = act_row_bytes(k, kind_blk);
    HIP_TRY(hipMalloc(&act, bytes));
    tview s; s.data = (char *) x; s.ne[0] = k; s.ne[1] = s.ne[2] = s.ne[3] = 1; s.nb[0] = 4; s.nb[1] = s.nb[2] = s.nb[3] = k * 4;
    int rc = launch_quantize_act(st, kind_blk, s, act, bytes);
    if (rc == CLLM_OK) {
        const int64_t nb = k / kind_blk;
        if (kind_blk == 32) hipLaunchKernelGGL(k_act_to_q8_0_blocks, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, (const char *) act, k, (block_q8_0 *) y);
        else                hipLaunchKernelGGL(k_act_to_q8_K_blocks, dim3((unsigned)((nb + 63) / 64)), dim3(64), 0, st, (const char *) act, k, (block_q8_K *) y);
        rc = cllm_hip_check(hipGetLastError(), "act_to_blocks", __FILE__, __LINE__);
    }
    (void) hipStreamSynchronize(st);
    (void) hipFree(act);
    return rc;
}

extern "C" int cllm_quantize_row_q8_0(void * stream, const float * x, void * y, int64_t k) { return quantize_to_blocks(stream, 32, x, y, k); }
extern "C" int cllm_quantize_row_q8_K(void * stream, const float * x, void * y, int64_t k) { return quantize_to_blocks(stream, 256, x, y, k); }
