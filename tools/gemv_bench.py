#!/usr/bin/env python3
"""GEMV kernel micro-benchmark (GPU box): HIP-event timed launches of the mat-mul kernel only, per shape/type.
usage: python tools/gemv_bench.py [--types q4_k,q4_0,q4_1,q8_0] [--cols 1] [--iters 64]
Tunables are read from the environment once per process (CLLM_MMVQ_WG, CLLM_MMVQ_OCC): sweep with --sweep."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

T = {"q4_k": 12, "q4_0": 2, "q4_1": 3, "q8_0": 8}
SHAPES = [("qkv", 4096, 6144), ("o", 4096, 4096), ("gate_up", 4096, 28672), ("down", 14336, 4096), ("lm_head", 4096, 128256),
          ("q72_down_q8", 29568, 8192)]


FUSED = [("qkv", 4096, 6144, 1, False), ("o", 4096, 4096, 2, True), ("gate_up", 4096, 28672, 1, False), ("gate_up_silu", 4096, 28672, 1, False),
         ("down", 14336, 4096, 3, True), ("down_q", 14336, 4096, 2, True),
         ("lm_head", 4096, 128256, 1, False)]


FUSED_Q72 = [("qkv", 8192, 10240, 1, False), ("o", 8192, 8192, 2, True), ("gate_up_silu", 8192, 59136, 1, False), ("down_q", 29568, 8192, 2, True),
             ("lm_head", 8192, 152064, 1, False)]       # Qwen2-72B (BASELINE cfg4): the one-GPU launches


def run_fused(types, iters, hot=False, force_pro=0, only="", shapes=None):
    """the decode launches: activation prologue (1 RMS_NORM+quantize, 2 quantize, 3 SiLU*up+quantize) inside the mat-vec"""
    pkg = ge.load_package()
    L = pkg.lib.get()
    pkg.lib.require_gpu()
    rng = np.random.default_rng(0)
    for tn in types:
        t = T[tn]
        for name, K, N, pro, resid in (shapes or FUSED):
            if only and name not in only.split(","):
                continue
            if force_pro and pro == 1:
                pro = force_pro
            nbytes = N * pkg.tensor.row_size(t, K)
            n_copies = 1 if hot else max(2, int(1.2 * 2**30 // nbytes) + 1)
            w0 = pkg.synth.make_tensor_fast("b." + name, t, N, K)
            ws = [pkg.Tensor.from_numpy(w0, t, [K, N]) for _ in range(n_copies)]
            x = pkg.Tensor.from_numpy(rng.standard_normal((1, K * (2 if pro == 3 else 1))).astype(np.float32))
            g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
            y = pkg.Tensor(pkg.F32, [N, 1])
            r = pkg.Tensor.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
            ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
            us = C.c_float()
            pkg.lib.check(L.cllm_bench_gemv_fused(None, t, ptrs, n_copies, K, N, pro, x.data_ptr(), g.data_ptr(), 1e-5, 1 if name == "gate_up_silu" else 0, y.data_ptr(),
                                                  r.data_ptr() if resid else None, iters, C.byref(us)), "bench")
            gbs = nbytes / (us.value * 1e-6) / 1e9
            print(f"fused {tn:5s} {name:9s} K={K:6d} N={N:6d} pro={pro} {nbytes/1e6:8.1f} MB {us.value:9.2f} us {gbs:8.1f} GB/s  {gbs/80:5.1f}% of 8TB/s  "
                  "", flush=True)
            del ws


def run(types, cols, iters, shapes, hot=False):
    pkg = ge.load_package()
    L = pkg.lib.get()
    pkg.lib.require_gpu()
    rng = np.random.default_rng(0)
    for tn in types:
        t = T[tn]
        for name, K, N in shapes:
            if K % pkg.tensor.BLCK[t]:
                continue
            nbytes = N * pkg.tensor.row_size(t, K)
            n_copies = 1 if hot else max(2, int(1.2 * 2**30 // nbytes) + 1)   # hot: same weights every launch (L2 / Infinity Cache resident)
            w0 = pkg.synth.make_tensor_fast("b." + name, t, N, K)
            ws = [pkg.Tensor.from_numpy(w0, t, [K, N]) for _ in range(n_copies)]
            x = pkg.Tensor.from_numpy(rng.standard_normal((cols, K)).astype(np.float32))
            y = pkg.Tensor(pkg.F32, [N, cols])
            cw, cx, cy = ws[0].c(), x.c(), y.c()
            wsize = L.cllm_mul_mat_wsize(C.byref(cw), C.byref(cx))
            scratch = pkg.tensor.Buffer(wsize + 256)
            ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
            us = C.c_float()
            pkg.lib.check(L.cllm_bench_mul_mat_kernel(None, C.byref(cw), ptrs, n_copies, C.byref(cx), C.byref(cy), scratch.ptr,
                                                      scratch.nbytes, iters, C.byref(us)), "bench")
            gbs = nbytes / (us.value * 1e-6) / 1e9
            print(f"{tn:5s} {name:12s} K={K:6d} N={N:6d} cols={cols} {nbytes/1e6:8.1f} MB {us.value:9.2f} us {gbs:8.1f} GB/s  "
                  f"{gbs/80:5.1f}% of 8TB/s  [wg={os.environ.get('CLLM_MMVQ_WG','256')} occ={os.environ.get('CLLM_MMVQ_OCC','8')}]", flush=True)
            del ws


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", default="q4_k,q4_0,q8_0")
    ap.add_argument("--cols", type=int, default=1)
    ap.add_argument("--iters", type=int, default=64)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--hot", action="store_true")
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--pro", type=int, default=0)
    ap.add_argument("--model", default="llama3-8b", choices=["llama3-8b", "qwen2-72b"], help="--fused: whose decode launches")
    a = ap.parse_args()
    shapes = [s for s in SHAPES if not a.shapes or s[0] in a.shapes.split(",")]
    if a.fused:
        run_fused(a.types.split(","), a.iters, a.hot, a.pro, a.shapes, FUSED_Q72 if a.model == "qwen2-72b" else None)
    elif a.sweep:
        for wg in (128, 256, 512):
            for occ in (4, 8, 16):
                env = dict(os.environ, CLLM_MMVQ_WG=str(wg), CLLM_MMVQ_OCC=str(occ))
                subprocess.call([sys.executable, __file__, "--types", a.types, "--cols", str(a.cols), "--iters", str(a.iters),
                                 "--shapes", a.shapes or "qkv,gate_up,down"], env=env)
    else:
        run(a.types.split(","), a.cols, a.iters, shapes, a.hot)
