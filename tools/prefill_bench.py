#!/usr/bin/env python3
"""Prefill throughput (BASELINE cfg3: Llama-3-8B shapes, Q4_0, 4096-token prompt) on one GPU: tokens/s of cllm_llama_forward
over the whole prompt + the algorithmic-FLOP fraction of the int8 MFMA peak (6.60e13 FLOP per 4096 tokens).
usage: python tools/prefill_bench.py [--model llama3-8b] [--wtype q4_0] [--n-prompt 4096] [--reps 3] [--layers N]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--wtype", default="q4_0", choices=sorted(bench.WTYPES))
    ap.add_argument("--n-prompt", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--layers", type=int, default=0, help="truncate the model to N layers (profiling)")
    a = ap.parse_args()
    pkg = ge.load_package()
    pkg.lib.require_gpu()
    wtype = bench.WTYPES[a.wtype]
    cfg = pkg.synth.config(a.model, max_len=(a.n_prompt + 63) // 64 * 64)
    if a.layers:
        cfg["n_layer"] = a.layers
    m = bench.build_model(pkg, cfg, wtype, 0, 1)
    prompt = np.random.default_rng(1234).integers(0, cfg["vocab"], a.n_prompt).astype(np.int32)
    m.forward(prompt, n_past=0)                     # warm-up (allocations, code objects)
    pkg.ops.sync()
    ts = []
    for _ in range(a.reps):
        t0 = time.perf_counter()
        m.forward(prompt, n_past=0)
        pkg.ops.sync()
        ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    H, hd, F, L = cfg["hidden"], cfg["head_dim"], cfg["ffn"], cfg["n_layer"]
    QD, KD = cfg["n_head"] * hd, cfg["n_kv_head"] * hd
    n = a.n_prompt
    lin = 2.0 * L * (H * (QD + 2 * KD) + QD * H + 3 * H * F) * n
    att = 2.0 * 2 * n * n * hd * cfg["n_head"] * L
    flops = lin + att
    print(f"prefill {a.model} {a.wtype} n={n} layers={L}: median {dt*1e3:.1f} ms  {n/dt:.0f} tok/s  "
          f"{flops/dt/1e12:.1f} TFLOP/s algorithmic ({lin/1e12:.1f} linear + {att/1e12:.1f} attention TFLOP) = {flops/dt/5e15*100:.1f}% of 5 PFLOP/s int8, "
          f"{flops/dt/2.5e15*100:.1f}% of 2.5 PFLOP/s f16")
