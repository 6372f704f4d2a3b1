#!/usr/bin/env python3
"""The decode step's FFN block (Llama-3-8B shapes, Q4_K) as the two launches of the five-launch layer against ONE fused launch (ffn_fused.hip) and against the fused launch
with a hand-off that costs nothing (the bound): HIP-event time per block over weight copies cycled past the Infinity Cache, then the fused launch's in-kernel stamps.
usage: python tools/ffn_bench.py [--iters 64] [--no-stamps]"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=64)
ap.add_argument("--hidden", type=int, default=4096)
ap.add_argument("--ffn", type=int, default=14336)
ap.add_argument("--no-stamps", action="store_true")
a = ap.parse_args()
pkg = ge.load_package()
L = pkg.lib.get()
pkg.lib.require_gpu()
L.cllm_debug_set_ffn_ts.argtypes = [C.c_void_p]
L.cllm_debug_set_ffn_ts.restype = None
H, F, t = a.hidden, a.ffn, 12
rng = np.random.default_rng(0)
per = (2 * F * pkg.tensor.row_size(t, H) + H * pkg.tensor.row_size(t, F))
n = max(2, int(1.5 * 2**30 // per) + 1)
wg0 = pkg.synth.make_tensor_fast("b.wgu", t, 2 * F, H)
wd0 = pkg.synth.make_tensor_fast("b.wd", t, H, F)
wgs = [pkg.Tensor.from_numpy(wg0, t, [H, 2 * F]) for _ in range(n)]
wds = [pkg.Tensor.from_numpy(wd0, t, [F, H]) for _ in range(n)]
pg = (C.c_void_p * n)(*[w.data_ptr().value for w in wgs])
pd = (C.c_void_p * n)(*[w.data_ptr().value for w in wds])
x0 = rng.standard_normal((1, H)).astype(np.float32)
nw = pkg.Tensor.from_numpy((1 + 0.1 * rng.standard_normal((1, H))).astype(np.float32))
g = pkg.Tensor(pkg.F32, [F, 1])
state = pkg.tensor.Buffer(L.cllm_ffn_fused_state_bytes(F))
L.cllm_memset(state.ptr, 0, L.cllm_ffn_fused_state_bytes(F), None)
res = {}
L.cllm_debug_set_ffn_flags.argtypes = [C.c_int]
L.cllm_debug_set_ffn_flags.restype = None
FLAGS = [int(f) for f in os.environ.get("FFN_BENCH_FLAGS", "1,0,3").split(",")]
for fl in FLAGS[1:]:
    L.cllm_debug_set_ffn_flags(fl)
    for name, mode in (("fused", 1), ("fused, null hand-off (bound)", 2)):
        x = pkg.Tensor.from_numpy(x0 * 0.0 + x0)
        us = C.c_float()
        pkg.lib.check(L.cllm_bench_ffn(None, pg, pd, n, H, F, x.data_ptr(), nw.data_ptr(), 1e-5, g.data_ptr(), state.ptr, mode, a.iters, C.byref(us)), "bench_ffn")
        print(f"flags {fl} (bit 0: ring filled behind the prologue barrier, bit 1: mid-stream gathers through the L2)  {name:32s} {us.value:7.2f} us per block", flush=True)
L.cllm_debug_set_ffn_flags(FLAGS[0])
print(f"flags {FLAGS[0]}:")
for name, mode in (("two launches", 0), ("fused", 1), ("fused, null hand-off (bound)", 2), ("two launches", 0), ("fused", 1)):
    x = pkg.Tensor.from_numpy(x0 * 0.0 + x0)          # x is updated in place: the same start for every form
    us = C.c_float()
    pkg.lib.check(L.cllm_bench_ffn(None, pg, pd, n, H, F, x.data_ptr(), nw.data_ptr(), 1e-5, g.data_ptr(), state.ptr, mode, a.iters, C.byref(us)), "bench_ffn")
    out = x.numpy().copy()
    res.setdefault(name, []).append((us.value, out))
    print(f"{name:32s} H={H} F={F}  {per / 1e6:6.1f} MB  {us.value:7.2f} us per block  {per / us.value / 1e3:7.1f} GB/s  ({100 * per / us.value / 1e3 / 8000:.1f} % of 8 TB/s)", flush=True)
same = np.array_equal(res["two launches"][0][1].view(np.uint32), res["fused"][0][1].view(np.uint32))
print("x after", a.iters + 4, "blocks: fused == two launches bit for bit:", same, " kernel errors:", L.cllm_check_kernel_errors())
if not a.no_stamps:
    ts = pkg.tensor.Buffer(256 * 8 * 8)
    for name, mode in (("fused", 1), ("fused, null hand-off (bound)", 2)):
        L.cllm_memset(ts.ptr, 0, 256 * 64, None)
        L.cllm_debug_set_ffn_ts(ts.ptr)
        x = pkg.Tensor.from_numpy(x0 * 0.0 + x0)
        us = C.c_float()
        pkg.lib.check(L.cllm_bench_ffn(None, pg, pd, n, H, F, x.data_ptr(), nw.data_ptr(), 1e-5, g.data_ptr(), state.ptr, mode, 16, C.byref(us)), "bench_ffn")
        L.cllm_debug_set_ffn_ts(C.c_void_p(0))
        host = np.zeros(256 * 8, dtype=np.uint64)
        pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h")
        L.cllm_stream_sync(None)
        st = host.reshape(256, 8).astype(np.int64)
        st = (st - st[:, 0].min()) / 100.0
        print(f"{name}: avg {us.value:.2f} us per block with stamps (thread 0 of every workgroup = wave 0)")
        for k, lab in enumerate(["entry", "first steps issued", "gate/up act row built", "prologue barrier", "gate/up units done", "edge: last block gathered", "edge barrier", "down rows done"]):
            c = st[:, k]
            print(f"    {k} {lab:26s} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f} us")
