#!/usr/bin/env python3
"""Sparse-MoE decode launches at Mixtral-8x7B shapes (GPU box): wall-clock per launch over many back-to-back calls (the launches are longer than the host's enqueue time).
  router            cllm_op_moe_router                        (RMS_NORM, router mat-vec, SOFT_MAX, TOP_K: one workgroup)
  gate_up           cllm_op_mul_mat_id_silu_mul               (two experts' gate / up rows, SiLU * up)
  router_gate_up    cllm_op_moe_router_gate_up                (both in one launch, round 5)
  down_combine      cllm_op_mul_mat_id_combine                (two experts' down rows, normalized weights, slot sum, residual)
Eight activation vectors are cycled so that the router picks different experts from call to call (8 experts x 99 MB: the weights do not stay in the 256 MB Infinity Cache).
usage: python tools/moe_bench.py [--iters 400]"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=400)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--ffn", type=int, default=14336)
    ap.add_argument("--experts", type=int, default=8)
    a = ap.parse_args()
    pkg = ge.load_package()
    L = pkg.lib.get()
    pkg.lib.require_gpu()
    ops, T = pkg.ops, pkg.Tensor
    K, F, E, k = a.hidden, a.ffn, a.experts, 2
    t = pkg.tensor.Q4_K
    rng = np.random.default_rng(0)
    packed = T.from_numpy(pkg.synth.make_tensor_fast("moe.gu", t, 2 * F * E, K), t, [K, 2 * F, E])
    down = T.from_numpy(pkg.synth.make_tensor_fast("moe.down", t, K * E, F), t, [F, K, E])
    gate_w = T.from_numpy(pkg.synth.make_tensor_fast("moe.router", t, E, K), t, [K, E])
    norm_w = T.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
    xs = [T.from_numpy(rng.standard_normal((1, K)).astype(np.float32)) for _ in range(8)]
    acts = [T.from_numpy((0.3 * rng.standard_normal((k, F))).astype(np.float32).reshape(1, k, F)) for _ in range(8)]
    resid = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
    xnorm = T(pkg.F32, [K]); probs = T(pkg.F32, [E, 1]); ids = T(pkg.I32, [k]); g = T(pkg.F32, [F, k, 1]); out = T(pkg.F32, [K, 1])
    ids2 = ids.view([k, 1], [4, 4 * k])
    R = ops._ref
    picks = set()
    for x in xs:
        pkg.lib.check(L.cllm_op_moe_router(None, R(x), R(norm_w), 1e-5, R(gate_w), R(xnorm), R(probs), R(ids)), "router")
        ops.sync()
        picks.add(tuple(int(v) for v in ids.numpy().reshape(-1)))
    print(f"expert pairs picked over the 8 activations: {sorted(picks)}", flush=True)
    xn_t = T(pkg.F32, [K, 1, 1])

    def timed(name, fn, nbytes):
        for i in range(16):
            fn(i)
        ops.sync()
        t0 = time.perf_counter()
        for i in range(a.iters):
            fn(i)
        ops.sync()
        us = (time.perf_counter() - t0) / a.iters * 1e6
        print(f"{name:16s} {us:8.2f} us" + (f"   {nbytes / us / 1e6:6.2f} TB/s of {nbytes / 1e6:.1f} MB" if nbytes else ""), flush=True)
        return us

    def f_router(i):
        pkg.lib.check(L.cllm_op_moe_router(None, R(xs[i % 8]), R(norm_w), 1e-5, R(gate_w), R(xn_t), R(probs), R(ids)), "router")

    def f_gate_up(i):      # (ids as the last router call left them; the activation is the normalized row)
        pkg.lib.check(L.cllm_op_mul_mat_id_silu_mul(None, R(packed), R(xn_t), R(ids2), R(g)), "gate_up")

    def f_both(i):
        f_router(i); f_gate_up(i)

    def f_fold(i):
        pkg.lib.check(L.cllm_op_moe_router_gate_up(None, R(xs[i % 8]), R(norm_w), 1e-5, R(gate_w), R(packed), R(probs), R(ids), R(g)), "router_gate_up")

    def f_down(i):
        pkg.lib.check(L.cllm_op_mul_mat_id_combine(None, R(down), R(acts[i % 8]), R(ids2), R(probs), R(resid), R(out)), "down_combine")

    def f_block(i):
        f_fold(i); f_down(i)

    gu_bytes = k * 2 * F * pkg.tensor.row_size(t, K)
    dn_bytes = k * K * pkg.tensor.row_size(t, F)
    timed("router", f_router, 0)
    timed("router+gate_up", f_both, gu_bytes)
    timed("router_gate_up", f_fold, gu_bytes)
    timed("down_combine", f_down, dn_bytes)
    timed("fold+down", f_block, gu_bytes + dn_bytes)


if __name__ == "__main__":
    main()
