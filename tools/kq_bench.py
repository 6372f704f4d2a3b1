"""Mat-vec timing of the Q5_K / Q6_K coverage kernel (gemv_kq.hip) at Llama-3-8B shapes, next to Q4_K through the same entry point (cllm_op_mul_mat).
usage: python tools/kq_bench.py"""
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
L = gpu.lib.get()
for tname, t in (("q4_k", gpu.Q4_K), ("q5_k", gpu.Q5_K), ("q6_k", gpu.Q6_K)):
    for name, N, K in (("qkv", 6144, 4096), ("down", 4096, 14336), ("gate_up", 28672, 4096), ("lm_head", 128256, 4096)):
        r = np.random.default_rng(1)
        w = gpu.synth.make_tensor_fast("w", t, N, K, 1) if hasattr(gpu.synth, "make_tensor_fast") else gpu.synth.make_tensor("w", t, N, K)
        ws = [gpu.Tensor.from_numpy(w, t, [K, N]) for _ in range(max(2, int(1.2e9 // w.nbytes)))]      # cycle through > 1.2 GB of copies
        x = gpu.Tensor.from_numpy(r.standard_normal((1, K)).astype(np.float32))
        dst = gpu.Tensor(gpu.F32, [N, 1])
        for i in range(3):
            gpu.ops.mul_mat(ws[i % len(ws)], x, dst)
        L.cllm_stream_sync(None)
        it = 24
        t0 = time.perf_counter()
        for i in range(it):
            gpu.ops.mul_mat(ws[i % len(ws)], x, dst)
        L.cllm_stream_sync(None)
        us = (time.perf_counter() - t0) / it * 1e6
        print(f"{tname} {name:8s} N={N:6d} K={K:5d}  {w.nbytes/1e6:7.1f} MB  {us:8.1f} us  {w.nbytes/us/1e6:6.2f} TB/s", flush=True)
        del ws
