#!/usr/bin/env python3
"""Where does a team-kernel launch (gemv_team32.hip) spend its time?  Per-wave wall-clock stamps (100 MHz): 0 entry, 1 prologue barrier passed, 2 the wave's loop done
(emit waves: last step handed over; chain waves: last unit stored).  Printed relative to the earliest entry of the launch."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

SHAPES = [("qkv", 4096, 6144, 1, False), ("o", 4096, 4096, 2, True), ("down_q", 14336, 4096, 2, True)]
pkg = ge.load_package()
L = pkg.lib.get()
pkg.lib.require_gpu()
lib = C.CDLL(pkg.lib.SO_PATH)
rng = np.random.default_rng(0)
NW = 256 * 16
ts = pkg.tensor.Buffer(NW * 4 * 8)
lib.cllm_debug_set_team32_ts.argtypes = [C.c_void_p]
for tname, t in (("q4_0", 2), ("q8_0", 8)):
    for name, K, N, pro, resid in SHAPES:
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
        L.cllm_memset(ts.ptr, 0, NW * 32, None)
        lib.cllm_debug_set_team32_ts(ts.ptr)
        pkg.lib.check(L.cllm_bench_gemv_fused(None, t, ptrs, n_copies, K, N, pro, x.data_ptr(), g.data_ptr(), 1e-5, 0, y.data_ptr(),
                                              r.data_ptr() if resid else None, 16, C.byref(us)), "bench")
        lib.cllm_debug_set_team32_ts(None)
        host = np.zeros(NW * 4, dtype=np.uint64)
        pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h")
        L.cllm_stream_sync(None)
        st = host.reshape(NW, 4).astype(np.int64)
        st = st[st[:, 0] > 0]
        if not len(st):
            print(f"{tname} {name:8s} K={K} N={N} pro={pro}  avg launch {us.value:.2f} us: not a team-kernel launch (the launcher's pick)")
            del ws
            continue
        base = st[:, 0].min()
        print(f"{tname} {name:8s} K={K} N={N} pro={pro}  avg launch {us.value:.2f} us (with stamps), {len(st)} waves stamped")
        for k, lab in enumerate(["entry", "prologue barrier", "loop done"]):
            c = (st[:, k][st[:, k] > 0] - base) / 100.0
            print(f"    {k} {lab:18s} min {c.min():6.2f}  median {np.median(c):6.2f}  p90 {np.percentile(c, 90):6.2f}  max {c.max():6.2f} us")
        nun = N // 8
        team = 16 if nun <= 256 else 8 if nun <= 512 else 5 if nun <= 768 else 4
        full = host.reshape(-1, 16, 4).astype(np.int64)
        chain_w = [q * team + ((q * (65 - team)) & 3) for q in range(16 // team)]
        for lab, sel in (("emit waves", [w for w in range((16 // team) * team) if w not in chain_w]), ("chain waves", chain_w)):
            a = full[:, sel, :].reshape(-1, 4)
            a = a[a[:, 2] > 0]
            d = (a[:, 2] - a[:, 1]) / 100.0
            e = (a[:, 2] - base) / 100.0
            mhz = np.median(a[:, 3] / np.maximum(d, 1e-3))
            print(f"      {lab:12s} shader clock during the loop {mhz:6.0f} MHz | loop duration min {d.min():6.2f}  median {np.median(d):6.2f}  p90 {np.percentile(d, 90):6.2f}  max {d.max():6.2f} | done at median {np.median(e):6.2f}  max {e.max():6.2f} us")
        del ws
