#!/usr/bin/env python3
"""The fused all-reduce's gather prologue (k_gemv_dec PRO 5, gemv_decode_kernel.h tpf_gather4) at N ranks: how long do the N ranks' granules of a 4096-value site take to arrive
in registers when they are all there already (no waiting: the scatters ran before on the same stream)?  Stamps of thread 0 of every workgroup: 6 = entry of the gather,
0 = gather done (activation in registers).  Compared with the plain RMS_NORM prologue's load (PRO 1) through the launch-to-launch time of the same mat-vec.
usage: python tools/tp_gather_probe.py"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
L = pkg.lib.get()
pkg.lib.require_gpu()
lib = C.CDLL(pkg.lib.SO_PATH)
lib.cllm_debug_set_tp_ts.argtypes = [C.c_void_p]
H, t = 4096, 12
rng = np.random.default_rng(0)
ts = pkg.tensor.Buffer(256 * 8 * 8)
for n in (1, 2, 4, 8):
    devs = (C.c_int * n)(*([0] * n))
    objs = (C.c_void_p * n)()
    pkg.lib.check(L.cllm_tp_fused_create_group(n, devs, 2, H, objs), "create_group")
    kr = H // n                                                      # every rank's K share of an o projection [K_r, H]
    wo = [pkg.Tensor.from_numpy(pkg.synth.make_tensor_fast(f"p.o{r}", t, H, kr), t, [kr, H]) for r in range(n)]
    att = [pkg.Tensor.from_numpy(rng.standard_normal((1, kr)).astype(np.float32)) for r in range(n)]
    wq = pkg.Tensor.from_numpy(pkg.synth.make_tensor_fast("p.qkv", t, 6144, H), t, [H, 6144])
    x = pkg.Tensor.from_numpy(rng.standard_normal((1, H)).astype(np.float32))
    g = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, H))).astype(np.float32))
    y = pkg.Tensor(pkg.F32, [6144, 1])
    xo = pkg.Tensor(pkg.F32, [H, 1])
    wo_c = [w.c() for w in wo]; wq_c = wq.c()
    durs, plain = [], []
    for it in range(12):
        for r in range(n):
            pkg.lib.check(L.cllm_tp_fused_advance(objs[r], None), "advance")
        for r in range(n):
            pkg.lib.check(L.cllm_op_mul_mat_vec_tp_scatter(None, C.byref(wo_c[r]), 2, att[r].data_ptr(), objs[r], 0), "scatter")
        L.cllm_stream_sync(None)
        L.cllm_memset(ts.ptr, 0, 256 * 64, None)
        lib.cllm_debug_set_tp_ts(ts.ptr)
        pkg.lib.check(L.cllm_op_mul_mat_vec_tp_gather(None, C.byref(wq_c), x.data_ptr(), g.data_ptr(), 1e-5, 0, None, y.data_ptr(), objs[0], 0, xo.data_ptr()), "gather")
        lib.cllm_debug_set_tp_ts(None)
        L.cllm_stream_sync(None)
        host = np.zeros(256 * 8, dtype=np.uint64)
        pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h")
        st = host.reshape(256, 8).astype(np.int64)
        ok = st[:, 6] > 0
        if it >= 2:
            durs.append(float(np.median((st[ok, 0] - st[ok, 6]) / 100.0)))
    print(f"{n} rank(s): gather of {H} values x {n} ranks' granules ({8 * H * n // 1024} KB per workgroup): median over the workgroups {np.median(durs):.2f} us (min {min(durs):.2f}, max {max(durs):.2f} over 10 launches)")
    for r in range(n):
        L.cllm_tp_fused_destroy(objs[r])
