#!/usr/bin/env python3
"""Where does a decode mat-vec launch spend its time?  Per-workgroup wall-clock stamps (100 MHz s_memrealtime) taken
inside k_mmvq_q4_K<.., FUSED>: 0 entry, 1 prologue + weight-prefetch loads issued, 2 activation row built (this wave),
3 prologue barrier passed, 4 this wave's rows done, 5 whole workgroup done.  Printed relative to the earliest entry."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SHAPES = [("qkv", 4096, 6144, 1, 0, False), ("o", 4096, 4096, 2, 0, True), ("gate_up_silu", 4096, 28672, 1, 1, False), ("down_q", 14336, 4096, 2, 0, True)]

pkg = ge.load_package()
L = pkg.lib.get()
pkg.lib.require_gpu()
lib = C.CDLL(pkg.lib.SO_PATH)
rng = np.random.default_rng(0)
ts = pkg.tensor.Buffer(256 * 8 * 8)
lib.cllm_debug_set_mmvq_ts.argtypes = [C.c_void_p]
set_ts = lib.cllm_debug_set_mmvq_ts
L.cllm_memset(ts.ptr, 0, 256 * 64, None)
for name, K, N, pro, epi, resid in SHAPES:
    t = 12
    nbytes = N * pkg.tensor.row_size(t, K)
    n_copies = max(2, int(1.2 * 2**30 // nbytes) + 1)
    w0 = pkg.synth.make_tensor_fast("b." + name, t, N, K)
    ws = [pkg.Tensor.from_numpy(w0, t, [K, N]) for _ in range(n_copies)]
    x = pkg.Tensor.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
    y = pkg.Tensor(pkg.F32, [N, 1])
    r = pkg.Tensor.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
    ptrs = (C.c_void_p * n_copies)(*[w.data_ptr().value for w in ws])
    us = C.c_float()
    set_ts(ts.ptr)
    pkg.lib.check(L.cllm_bench_gemv_fused(None, t, ptrs, n_copies, K, N, pro, x.data_ptr(), g.data_ptr(), 1e-5, epi, y.data_ptr(),
                                          r.data_ptr() if resid else None, 16, C.byref(us)), "bench")
    set_ts(None)
    host = np.zeros(256 * 8, dtype=np.uint64)
    pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h")
    L.cllm_stream_sync(None)
    st = host.reshape(256, 8)[:, :6].astype(np.int64)
    st = (st - st[:, 0].min()) / 100.0       # us
    print(f"{name:13s} K={K} N={N} pro={pro}  avg launch {us.value:.2f} us (with stamps)")
    for k, lab in enumerate(["entry", "loads issued", "act row built", "prologue barrier", "rows done (wave 0)", "workgroup done"]):
        c = st[:, k]
        print(f"    {k} {lab:20s} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f} us")
    del ws
