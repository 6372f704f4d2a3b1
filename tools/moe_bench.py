#!/usr/bin/env python3
"""BASELINE cfg5 shapes (Mixtral-8x7B, Q4_K): the decode step's expert mat-vecs through GGML_OP_MUL_MAT_ID
(cllm_op_mul_mat_id: 8 experts resident, 2 routed per token), HIP-event timed, the routed pair changing every launch.
Per layer a token touches 2 x (gate 33 MB + up 33 MB + down 33 MB) of the 792 MB of expert weights."""
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
E, U, H, F, t = 8, 2, 4096, 14336, 12
rng = np.random.default_rng(0)
pairs = [rng.choice(E, U, replace=False).astype(np.int32).reshape(1, U) for _ in range(32)]
ids_t = [pkg.Tensor.from_numpy(p, pkg.I32) for p in pairs]
ptrs = (C.c_void_p * len(ids_t))(*[i.data_ptr().value for i in ids_t])
tot_us, tot_bytes = 0.0, 0
for name, K, N in (("gate_exps", H, F), ("up_exps", H, F), ("down_exps", F, H)):
    rb = pkg.tensor.row_size(t, K)
    w = np.concatenate([pkg.synth.make_tensor_fast(f"moe.{name}.{e}", t, N, K) for e in range(E)], axis=0)
    as_ = pkg.Tensor.from_numpy(w, t, [K, N, E])
    b = pkg.Tensor.from_numpy(rng.standard_normal((1, 1 if name != "down_exps" else U, K)).astype(np.float32))      # [K, 1 or U, 1 token]
    dst = pkg.Tensor(pkg.F32, [N, U, 1, 1])
    ca, cb, ci, cd = as_.c(), b.c(), ids_t[0].c(), dst.c()
    ws = L.cllm_mul_mat_wsize(C.byref(ca), C.byref(cb))
    scratch = pkg.tensor.Buffer(ws + 256)
    us = C.c_float()
    pkg.lib.check(L.cllm_bench_mul_mat_id(None, C.byref(ca), C.byref(cb), C.byref(ci), ptrs, len(ids_t), C.byref(cd), scratch.ptr, scratch.nbytes, 64, C.byref(us)), "bench")
    nbytes = U * N * rb
    tot_us += us.value; tot_bytes += nbytes
    print(f"mul_mat_id {name:10s} K={K:6d} N={N:6d} experts {E} routed {U}: {us.value:7.2f} us  {nbytes/1e6:6.1f} MB touched  {nbytes/us.value/1e3:7.1f} GB/s")
    del as_
print(f"expert mat-vecs of one layer: {tot_us:.1f} us for {tot_bytes/1e6:.0f} MB = {tot_bytes/tot_us/1e3:.0f} GB/s; x32 layers = {32*tot_us/1e3:.2f} ms/token of expert work")
