"""Timing of the flash-attention kernel (GPU): prefill (eager node sequence vs the fused kernel vs FLASH_ATTN_EXT with a mask tensor)
and decode at several context lengths (F16 and Q8_0 caches).  llama3-8b head layout: 32 heads, 8 kv heads, head size 128.
usage: python tools/fattn_bench.py [N_prefill]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package  # noqa: E402

gpu = load_package()
gpu.lib.get()
gpu.lib.require_gpu()
T, ops, L = gpu.Tensor, gpu.ops, gpu.lib.get()
D, H, Hkv = 128, 32, 8


def timeit(fn, iters=10):
    fn()
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    gpu.lib.check(L.cllm_stream_sync(None), "sync")
    return (time.perf_counter() - t0) / iters


def prefill(N):
    r = np.random.default_rng(1)
    ML, KD = int(os.environ.get("FA_ML", N)), D * Hkv      # FA_ML: rows of the transposed V cache (a power of two camps on few L2 channels)
    q = T.from_numpy(r.standard_normal((N, H, D)).astype(np.float32), gpu.F32, [D, H, N]).permute(0, 2, 1, 3)
    kc = T.from_numpy((r.standard_normal((ML, KD)) * 0.5).astype(np.float16), gpu.F16, [KD, ML])
    vc = T.from_numpy(r.standard_normal((KD, ML)).astype(np.float16), gpu.F16, [ML, KD])
    k = kc.view([D, N, Hkv], [2, KD * 2, D * 2])
    vt = vc.view([N, D, Hkv], [2, ML * 2, ML * D * 2])
    scale = 1.0 / np.sqrt(D)
    flops = 4.0 * N * N * D * H / 2
    dst = T(gpu.F32, [D, N, H])
    t = timeit(lambda: ops.attn_prefill(q, k, vt, scale, 0, dst))
    print(f"prefill N={N}: fused causal kernel      {t*1e3:8.3f} ms  {flops/t/1e12:7.1f} TFLOP/s (causal flops)")
    if os.environ.get("FA_ONLY_FUSED"):
        return
    # FLASH_ATTN_EXT with the mask tensor the host uploads; V rows by position
    v = T.from_numpy(r.standard_normal((Hkv, N, D)).astype(np.float16), gpu.F16, [D, N, Hkv])
    m = np.zeros((N, N), np.float16)
    m[np.triu_indices(N, 1)] = -np.inf
    dm = T.from_numpy(m, gpu.F16, [N, N])
    dst2 = T(gpu.F32, [D, H, N])
    t = timeit(lambda: ops.flash_attention(q, k, v, dm, scale, dst=dst2))
    print(f"prefill N={N}: FLASH_ATTN_EXT (mask)    {t*1e3:8.3f} ms  {flops/t/1e12:7.1f} TFLOP/s")
    # the node sequence it replaces
    S = T(gpu.F32, [N, N, H])
    ctx = T(gpu.F32, [D, N, H])

    def eager():
        ops.mul_mat(k, q, S)
        L.cllm_op_scale_mask_soft_max(None, S.c(), S.c(), scale, 0)
        ops.mul_mat(vt, S, ctx)
    try:
        t = timeit(eager, 3)
        print(f"prefill N={N}: MUL_MAT+SOFT_MAX+MUL_MAT {t*1e3:8.3f} ms  {flops/t/1e12:7.1f} TFLOP/s")
    except Exception as e:  # noqa: BLE001
        print("eager path failed:", e)


def decode(n_kv, kv_t):
    r = np.random.default_rng(2)
    q = T.from_numpy(r.standard_normal((H, 1, D)).astype(np.float32), gpu.F32, [D, 1, H])
    if kv_t == gpu.F16:
        k = T.from_numpy((r.standard_normal((Hkv, n_kv, D)) * 0.5).astype(np.float16), gpu.F16, [D, n_kv, Hkv])
        v = T.from_numpy(r.standard_normal((Hkv, n_kv, D)).astype(np.float16), gpu.F16, [D, n_kv, Hkv])
        bpp = 2 * D * 2
    else:
        blocks = r.integers(0, 255, (Hkv * n_kv * (D // 32) * 34), dtype=np.uint8)
        blocks.reshape(-1, 34)[:, 0:2] = np.frombuffer(np.float16(0.01).tobytes(), np.uint8)
        k = T.from_numpy(blocks, gpu.Q8_0, [D, n_kv, Hkv])
        v = T.from_numpy(blocks, gpu.Q8_0, [D, n_kv, Hkv])
        bpp = 2 * (D // 32) * 34
    m = T.from_numpy(np.zeros((1, n_kv), np.float16), gpu.F16, [n_kv, 1])
    dst = T(gpu.F32, [D, H, 1])
    t = timeit(lambda: ops.flash_attention(q, k, v, m, 1.0 / np.sqrt(D), dst=dst), 50)
    gb = n_kv * Hkv * bpp / 1e9
    print(f"decode n_kv={n_kv:6d} {'F16 ' if kv_t == gpu.F16 else 'Q8_0'}: {t*1e6:8.1f} us   cache {gb*1e3:7.2f} MB -> {gb/t/1e3:6.2f} TB/s")


if __name__ == "__main__":
    for n in ([int(sys.argv[1])] if len(sys.argv) > 1 else [512, 4096]):
        prefill(n)
    if os.environ.get("FA_ONLY_FUSED"):
        sys.exit(0)
    for n_kv in (300, 1024, 4096, 16384):
        decode(n_kv, gpu.F16)
    for n_kv in (4096, 16384):
        decode(n_kv, gpu.Q8_0)
