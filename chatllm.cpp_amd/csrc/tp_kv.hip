// tp_kv.hip -- the KV-cache shards of the ggml module's logical tensor-parallel device (chatllm.cpp_amd/host/ggml-hip.cpp).
//
// The host owns ONE cache per layer (KVCacheAttention, src/layers.cpp:3044-3123: K rows [n_ctx][kvH * hd] F16, V transposed [kvH * hd][max_len] F16) in rank 0's memory: prompts
// run there un-sharded and write it.  A tensor-parallel decode step attends, on rank r, over the heads of that rank only, out of a dense shard in rank r's own HBM
// (K [max_len][KD_r], V [KD_r][max_len]); these kernels move rows between the two: after a prompt the shard is refreshed from the host's cache (rows [p0, p1)), after every
// tensor-parallel step the one new row goes back (position read on the device: the launch stays valid inside a captured step), so the host's cache is coherent whenever the
// host -- or an un-sharded graph -- looks at it.  Byte copies: no arithmetic, nothing to deviate.
#include "common.h"

struct kv_shard_entry { uint16_t * auth_k, * auth_v, * shard_k, * shard_v; };      // one per layer (device table built by the caller)

template <bool TO_AUTH>
__global__ void __launch_bounds__(256) k_kv_shard_copy(const kv_shard_entry * __restrict__ tab, int KDr, int KD, int off, int ML, int p0, int p1, const int32_t * __restrict__ pos_dev) {
    const kv_shard_entry e = tab[blockIdx.y];
    if (pos_dev) { p0 = *pos_dev; p1 = p0 + 1; }
    const long total = (long)(p1 - p0) * KDr;
    for (long i = (long) blockIdx.x * 256 + threadIdx.x; i < total; i += (long) gridDim.x * 256) {
        const int p = p0 + (int)(i / KDr), c = (int)(i % KDr);
        if (TO_AUTH) {
            e.auth_k[(long) p * KD + off + c] = e.shard_k[(long) p * KDr + c];
            e.auth_v[(long)(off + c) * ML + p] = e.shard_v[(long) c * ML + p];
        } else {
            e.shard_k[(long) p * KDr + c] = e.auth_k[(long) p * KD + off + c];
            e.shard_v[(long) c * ML + p] = e.auth_v[(long)(off + c) * ML + p];
        }
    }
}

// table_dev: n_layers x { authoritative K, authoritative V, shard K, shard V } (device pointers, in device memory of the launching GPU).  kd_shard values per row of the shard =
// columns [kd_offset, kd_offset + kd_shard) of the kd_full-wide authoritative rows.  pos_dev != NULL: the one row at *pos_dev (p0 / p1 ignored); else rows [p0, p1).
extern "C" CLLM_API int cllm_op_kv_shard_copy(void * stream, const void * table_dev, int n_layers, int kd_shard, int kd_full, int kd_offset, int64_t max_len, int64_t p0, int64_t p1,
                                              const int32_t * pos_dev, int to_authoritative) {
    if (!table_dev || n_layers <= 0 || kd_shard <= 0 || kd_full < kd_shard || kd_offset < 0 || kd_offset + kd_shard > kd_full || max_len <= 0 || max_len > INT32_MAX)
        FAIL(CLLM_E_INVALID, "kv_shard_copy: arguments");
    if (!pos_dev && (p0 < 0 || p1 < p0 || p1 > max_len)) FAIL(CLLM_E_INVALID, "kv_shard_copy: rows [%lld, %lld) of %lld", (long long) p0, (long long) p1, (long long) max_len);
    if (!pos_dev && p0 == p1) return CLLM_OK;
    const long total = pos_dev ? kd_shard : (long)(p1 - p0) * kd_shard;
    long gx = (total + 255) / 256; if (gx > 1024) gx = 1024;
    const dim3 grid((unsigned) gx, (unsigned) n_layers);
    if (to_authoritative) hipLaunchKernelGGL(k_kv_shard_copy<true>,  grid, dim3(256), 0, (hipStream_t) stream, (const kv_shard_entry *) table_dev, kd_shard, kd_full, kd_offset, (int) max_len, (int) p0, (int) p1, pos_dev);
    else                  hipLaunchKernelGGL(k_kv_shard_copy<false>, grid, dim3(256), 0, (hipStream_t) stream, (const kv_shard_entry *) table_dev, kd_shard, kd_full, kd_offset, (int) max_len, (int) p0, (int) p1, pos_dev);
    LAUNCH_CHECK();
    return CLLM_OK;
}

// a pitched device-to-device copy (the K-split of a quantized weight matrix: whole quant blocks [b0, b1) of every row; a column range of any row-major matrix)
extern "C" CLLM_API int cllm_copy_2d(void * stream, void * dst, size_t dst_pitch, const void * src, size_t src_pitch, size_t width_bytes, size_t rows) {
    if (!dst || !src || width_bytes > dst_pitch || width_bytes > src_pitch) FAIL(CLLM_E_INVALID, "copy_2d: arguments");
    if (!width_bytes || !rows) return CLLM_OK;
    HIP_TRY(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, (hipStream_t) stream));
    return CLLM_OK;
}
