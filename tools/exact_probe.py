"""Bit-exactness probe (GPU): every op of the path that should now equal the oracle (== the reference build) to the last bit.
Prints mismatch counts instead of asserting, so that one GPU run shows everything.  usage: python tools/exact_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_package  # noqa: E402
import oracle as O  # noqa: E402
from synth_helpers import rand_blocks  # noqa: E402

gpu = load_package()
gpu.lib.get()
gpu.lib.require_gpu()
rng = np.random.default_rng(123)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def report(name, got, want):
    got, want = np.asarray(got, np.float32).ravel(), np.asarray(want, np.float32).ravel()
    bad = int(np.sum(bits(got) != bits(want)))
    rel = float(np.max(np.abs(got.astype(np.float64) - want)) / (np.max(np.abs(want)) + 1e-30))
    print(f"{'OK  ' if bad == 0 else 'DIFF'} {name}: {bad}/{want.size} words differ, max rel {rel:.2e}", flush=True)
    if bad and bad > 4:
        print('     first got', got[:6].tolist(), 'want', want[:6].tolist())
    if bad and bad <= 4:
        i = np.nonzero(bits(got) != bits(want))[0]
        print("     idx", i.tolist(), "got", got[i].tolist(), "want", want[i].tolist())
    return bad


def mm(t, K, N, M):
    w = rand_blocks(t, N, K, rng)
    x = rng.standard_normal((M, K)).astype(np.float32)
    want = np.zeros((M, N), np.float32)
    O.mul_mat(O.tensor(w, t, [K, N]), O.tensor(x, O.F32, [K, M]), O.tensor(want, O.F32, [N, M]))
    got = gpu.ops.mul_mat(gpu.Tensor.from_numpy(w, t, [K, N]), gpu.Tensor.from_numpy(x)).numpy()
    return got, want


def main():
    names = {O.Q4_K: "q4_K", O.Q4_0: "q4_0", O.Q8_0: "q8_0", O.Q4_1: "q4_1"}
    total = 0
    # ---- mat-vec, 1..8 columns (mmvq.hip) ----
    for t in (O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1):
        for K, N, M in ((256, 8, 1), (512, 7, 1), (2048, 64, 1), (4096, 130, 1), (14336, 40, 1), (2304, 33, 1), (1024, 33, 3), (768, 40, 4), (1280, 24, 8), (4096, 64, 2)):
            if t == O.Q4_K and K % 256:
                continue
            got, want = mm(t, K, N, M)
            total += report(f"mul_mat {names[t]} K={K} N={N} M={M}", got, want)
    # ---- the decode mat-vec with prologues (gemv_decode.hip): norm / plain / SiLU prologues, residual epilogue ----
    L = gpu.lib.get()
    for t in (O.Q4_K, O.Q4_0, O.Q8_0, O.Q4_1):
        for K, N in ((4096, 6144), (4096, 512), (14336, 256), (2048, 100), (4096, 16384), (1024, 64), (8192, 1032)):
            w = rand_blocks(t, N, K, rng)
            x = rng.standard_normal(K).astype(np.float32)
            nw = (1 + 0.1 * rng.standard_normal(K)).astype(np.float32)
            resid = rng.standard_normal(N).astype(np.float32)
            # oracle: rms_norm * w -> mul_mat (+ resid)
            xn = np.zeros_like(x)
            O.rms_norm(O.tensor(x, O.F32, [K]), O.tensor(xn, O.F32, [K]), 1e-5)
            xn = (xn * nw).astype(np.float32)
            want = np.zeros(N, np.float32)
            O.mul_mat(O.tensor(w, t, [K, N]), O.tensor(xn, O.F32, [K, 1]), O.tensor(want, O.F32, [N, 1]))
            want = (want + resid).astype(np.float32)
            dw = gpu.Tensor.from_numpy(w, t, [K, N])
            import ctypes as C
            dx, dn, dr, out = gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(nw), gpu.Tensor.from_numpy(resid), gpu.Tensor(gpu.F32, [N, 1])
            cw = dw.c()
            gpu.lib.check(L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), 1, dx.data_ptr(), dn.data_ptr(), 1e-5, 0, dr.data_ptr(), out.data_ptr()), "fused")
            total += report(f"gemv_decode {names[t]} pro1 K={K} N={N}", out.numpy(), want)
    # ---- rope (glibc cos/sin + the reference's fma form) ----
    for mode in (0, 2):
        hd, heads, qlen = 128, 8, 5
        x = rng.standard_normal((qlen, heads, hd)).astype(np.float32)
        pos = np.array([0, 3, 121, 4097, 70000], np.int32)
        want = np.zeros_like(x)
        O.rope(O.tensor(x, O.F32, [hd, heads, qlen]), pos, None, O.tensor(want, O.F32, [hd, heads, qlen]), hd, mode, 500000.0)
        got = gpu.ops.rope_ext(gpu.Tensor.from_numpy(x), gpu.Tensor.from_numpy(pos), None, hd, mode, freq_base=500000.0).numpy()
        total += report(f"rope mode {mode}", got, want)
    # ---- soft_max / silu with n % 8 tails (glibc expf) ----
    for n in (77, 256, 1001):
        x = (rng.standard_normal((4, n)) * 4).astype(np.float32)
        want = np.zeros_like(x)
        O.soft_max(O.tensor(x, O.F32, [n, 4]), None, O.tensor(want, O.F32, [n, 4]))
        total += report(f"soft_max n={n}", gpu.ops.soft_max(gpu.Tensor.from_numpy(x)).numpy(), want)
        want = np.zeros_like(x)
        O.silu(O.tensor(x, O.F32, [n, 4]), O.tensor(want, O.F32, [n, 4]))
        total += report(f"silu n={n}", gpu.ops.silu(gpu.Tensor.from_numpy(x)).numpy(), want)
    # ---- rms_norm ----
    for n in (4096, 8192, 100):
        x = (rng.standard_normal((3, n)) * 2).astype(np.float32)
        want = np.zeros_like(x)
        O.rms_norm(O.tensor(x, O.F32, [n, 3]), O.tensor(want, O.F32, [n, 3]), 1e-5)
        total += report(f"rms_norm n={n}", gpu.ops.rms_norm(gpu.Tensor.from_numpy(x), 1e-5).numpy(), want)
    print("TOTAL words differing:", total)


if __name__ == "__main__":
    main()
