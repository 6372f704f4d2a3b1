#!/usr/bin/env python3
"""Randomized differential run of the C ABI against the oracle (== libggml-cpu.so bits): random shapes for MUL_MAT of every weight type and column count, the fused
decode mat-vec forms, MUL_MAT_ID, the single-token attention block at random context lengths, and the device weight quantizers.  Everything must be BIT-identical.
usage (GPU box): python tools/fuzz_parity.py [--seconds 120] [--seed 1]      prints one line per mismatch and a summary; exit code 1 on any mismatch"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import __graft_entry__ as ge  # noqa: E402
import oracle as O  # noqa: E402
from synth_helpers import rand_blocks  # noqa: E402

QT = [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K, O.Q2_K, O.Q3_K, O.Q5_0, O.Q5_1, O.IQ4_NL, O.MXFP4, O.IQ4_XS, O.TQ1_0, O.TQ2_0, O.IQ2_XXS, O.IQ2_XS, O.IQ2_S, O.IQ3_XXS, O.IQ3_S, O.IQ1_S, O.IQ1_M]
TUNED = [O.Q4_0, O.Q4_1, O.Q8_0, O.Q4_K]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    gpu = ge.load_package()
    gpu.lib.require_gpu()
    L, T, ops = gpu.lib.get(), gpu.Tensor, gpu.ops
    rng = np.random.default_rng(a.seed)
    bad, n_cases = [], {}
    t_end = time.time() + a.seconds

    def note(kind, ok, desc):
        n_cases[kind] = n_cases.get(kind, 0) + 1
        if not ok:
            bad.append((kind, desc))
            print("MISMATCH", kind, desc, flush=True)

    def mul_mat():
        t = int(rng.choice(QT + [O.F16]))
        blk = 1 if t == O.F16 else O.BLCK[t]
        K = int(rng.choice([1, 2, 3, 4, 8, 16, 17, 44, 56, 115])) * max(blk, 32) if rng.random() < 0.8 else blk * int(rng.integers(1, 40))
        if t == O.F16:
            K = int(rng.choice([64, 96, 128, 100, 1000, 1001, 77, 256]))
        N = int(rng.choice([1, 2, 7, 8, 16, 24, 33, 64, 100, 130, 256, 1000, 2056, 17000 if K <= 1024 else 520]))
        M = int(rng.choice([1, 1, 1, 2, 3, 4, 5, 8, 9, 10, 12, 16, 31, 32, 33, 40, 64, 65, 100]))
        if t == O.F16:
            w = rng.standard_normal((1, N, K)).astype(np.float16)
        else:
            w = rand_blocks(t, N, K, rng)
        x = (rng.standard_normal((1, M, K)) * rng.choice([0.05, 1.0, 20.0])).astype(np.float32)
        want = np.zeros((1, M, N), np.float32)
        O.mul_mat(O.tensor(w, t, [K, N, 1]), O.tensor(x, O.F32, [K, M, 1]), O.tensor(want, O.F32, [N, M, 1]))
        got = ops.mul_mat(T.from_numpy(w, t, [K, N, 1]), T.from_numpy(x)).numpy()
        note("mul_mat", np.array_equal(got.view(np.uint32).ravel(), want.view(np.uint32).ravel()), f"type {t} K {K} N {N} M {M}")

    def fused():
        t = int(rng.choice(TUNED))
        K = int(rng.choice([256, 512, 1024, 2048, 4096, 8192, 14336]))
        N = int(rng.choice([8, 16, 64, 96, 200, 384, 1024, 4112, 18000 if K <= 1024 else 264]))
        pro = int(rng.choice([1, 2, 4]))
        w = T.from_numpy(rand_blocks(t, N, K, rng), t, [K, N])
        x = T.from_numpy(rng.standard_normal((1, K)).astype(np.float32))
        g = T.from_numpy((1 + 0.1 * rng.standard_normal((1, K))).astype(np.float32))
        r = T.from_numpy(rng.standard_normal((1, N)).astype(np.float32))
        act = ops.rms_norm_mul(x, T.from_numpy(g.numpy().reshape(K)), 1e-5) if pro == 1 else ops.silu_mul(x, g) if pro == 4 else x
        L.cllm_debug_set_gemv_rows32(0)
        L.cllm_debug_set_gemv_team32(0)
        want = ops.add(ops.mul_mat(w, act), r).numpy()
        L.cllm_debug_set_gemv_rows32(int(rng.choice([1, 1, 2, 4, 8])))
        L.cllm_debug_set_gemv_team32(int(rng.choice([1, 1, 4, 5, 8, 16])))
        out = T(gpu.F32, [N, 1])
        cw = w.c()
        rc = L.cllm_op_mul_mat_vec_fused(None, C.byref(cw), pro, x.data_ptr(), g.data_ptr() if pro != 2 else None, 1e-5, 0, r.data_ptr(), out.data_ptr())
        L.cllm_debug_set_gemv_rows32(1)
        L.cllm_debug_set_gemv_team32(1)
        note("fused mat-vec", rc == 0 and np.array_equal(out.numpy().view(np.uint32).ravel(), want.view(np.uint32).ravel()), f"type {t} K {K} N {N} pro {pro} rc {rc}")

    def mul_mat_id():
        t = int(rng.choice(QT))
        K = O.BLCK[t] * int(rng.integers(1, 5)) if O.BLCK[t] == 256 else 32 * int(rng.choice([2, 3, 8, 16, 33]))
        N, E, U, Tk = int(rng.choice([8, 40, 100])), int(rng.choice([4, 8])), int(rng.choice([1, 2, 3])), int(rng.choice([1, 1, 2, 5]))
        w = rand_blocks(t, N * E, K, rng)
        nb1 = int(rng.choice([1, U]))
        x = rng.standard_normal((Tk, nb1, K)).astype(np.float32)
        ids = rng.integers(0, E, (Tk, U)).astype(np.int32)
        want = np.zeros((Tk, U, N), np.float32)
        O.mul_mat_id(O.tensor(w, t, [K, N, E]), O.tensor(x, O.F32, [K, nb1, Tk]), O.tensor(ids, O.I32, [U, Tk]), O.tensor(want, O.F32, [N, U, Tk]))
        got = ops.mul_mat_id(T.from_numpy(w, t, [K, N, E]), T.from_numpy(x), T.from_numpy(ids)).numpy()
        note("mul_mat_id", np.array_equal(got.reshape(want.shape).view(np.uint32), want.view(np.uint32)), f"type {t} K {K} N {N} E {E} U {U} T {Tk} nb1 {nb1}")

    def attention():
        hd = int(rng.choice([64, 128]))
        nkv = int(rng.choice([1, 2, 4, 8]))
        nh = nkv * int(rng.choice([1, 2, 4, 8]))
        ML = int(rng.choice([256, 1024, 2048, 4096, 8192]))
        n_past = int(rng.integers(0, ML - 1))
        mode = int(rng.choice([0, 2]))
        QD, KD, n_kv, fb = hd * nh, hd * nkv, n_past + 1, 500000.0
        qkv = rng.standard_normal(QD + 2 * KD).astype(np.float32)
        kc0 = rng.standard_normal((ML, KD)).astype(np.float16)
        vc0 = rng.standard_normal((KD, ML)).astype(np.float16)
        pos = T.from_numpy(np.array([n_past], np.int32))
        dk, dv = T.from_numpy(kc0), T.from_numpy(vc0)
        q = T.from_numpy(qkv[:QD].reshape(1, nh, hd).copy())
        k = T.from_numpy(qkv[QD:QD + KD].reshape(1, nkv, hd).copy())
        v = T.from_numpy(qkv[QD + KD:].reshape(1, KD).copy())
        ops.cpy(v.transpose(), dv.view([1, KD], [2, ML * 2], offset=n_past * 2))
        kr = ops.rope_ext(k, pos, None, hd, mode, freq_base=fb, inplace=True)
        ops.set_rows(dk.view([KD, ML], [2, KD * 2]), kr.reshape(KD, 1), pos)
        qr = ops.rope_ext(q, pos, None, hd, mode, freq_base=fb, inplace=True)
        s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD * 2, hd * 2]), qr.permute(0, 2, 1, 3))
        p = ops.scale_mask_soft_max(s, float(np.float32(1.0) / np.sqrt(np.float32(hd))), n_past)
        c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML * 2, ML * hd * 2]), p)
        want = ops.cont(c.permute(0, 2, 1, 3)).numpy().reshape(QD)
        fk, fv = T.from_numpy(kc0), T.from_numpy(vc0)
        got = ops.rope_kv_attn_decode(T.from_numpy(qkv), pos, n_kv, nh, nkv, hd, mode, fb, fk, fv, ML, table=True).numpy().reshape(QD)
        ok = np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.array_equal(fk.numpy().view(np.uint16), dk.numpy().view(np.uint16)) and \
            np.array_equal(fv.numpy().view(np.uint16), dv.numpy().view(np.uint16))
        note("attention block", ok, f"hd {hd} nh {nh} nkv {nkv} ML {ML} n_past {n_past} mode {mode}")

    def quantize():
        t = int(rng.choice([O.Q8_0, O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.Q4_K, O.F16]))
        K, rows = 256 * int(rng.integers(1, 9)), int(rng.integers(1, 9))
        x = (rng.standard_normal((rows, K)) * rng.choice([1e-3, 1.0, 60.0])).astype(np.float32)
        if rng.random() < 0.3:
            x[0, : K // 2] = np.round(x[0, : K // 2] * 8) / 8
        want = np.concatenate([O.quantize_ref(t, x[r]) for r in range(rows)])
        out = T(t, [K, rows])
        rc = L.cllm_op_quantize_rows(None, t, T.from_numpy(x).data_ptr(), out.data_ptr(), K, rows)
        note("weight quantizer", rc == 0 and np.array_equal(out.numpy().view(np.uint8).reshape(-1), want), f"type {t} K {K} rows {rows}")

    kinds = [mul_mat, mul_mat, mul_mat, fused, mul_mat_id, attention, quantize]
    while time.time() < t_end:
        kinds[int(rng.integers(0, len(kinds)))]()
    print("cases:", ", ".join(f"{k} {v}" for k, v in sorted(n_cases.items())), "| mismatches:", len(bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
