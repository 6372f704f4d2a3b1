#!/usr/bin/env python3
"""In-kernel wall-clock stamps of the persistent all-layers decode launch (decode_persist.hip, workgroup 0): per phase, the time from the phase's start
(weight prefetch issued, then the wait on the previous phase's barrier) to its publish, averaged over the layers of the LAST decoded token.
usage: CLLM_PERSIST_TS=1 python tools/persist_phase_probe.py [--model llama3-8b] [--steps 40]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("CLLM_PERSIST_TS", "1")
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--steps", type=int, default=40)
    a = ap.parse_args()
    pkg = ge.load_package()
    pkg.lib.require_gpu()
    cfg = pkg.synth.config(a.model, max_len=256)
    m = bench.build_model(pkg, cfg, bench.WTYPES["q4_k"], 0, 1)
    prompt = np.random.default_rng(1234).integers(0, cfg["vocab"], 16).astype(np.int32)
    tok = int(np.argmax(m.forward(prompt)))
    m.decode_greedy(tok, a.steps)
    L = cfg["n_layer"]
    ts = m.debug_read("persist_ts", L * 40 * 2).view(np.uint64).reshape(L, 5, 8).astype(np.float64) * 0.01      # wall_clock64: 100 MHz -> us
    names = ["qkv (norm+quant+mat-vec)", "attention", "o (+residual)", "gate/up (+SiLU*up)", "down (+residual)"]
    dur = ts[:, :, 1] - ts[:, :, 0]
    print(f"{a.model} Q4_K, persistent decode launch, workgroup 0, last token: us per phase, mean over layers 1..{L - 1} (min .. max) | prefetch issue + barrier wait, activation prologue, row loop, publish")
    for i, n in enumerate(names):
        d = dur[1:, i]
        parts = ""
        if i != 1:
            t = ts[1:, i]
            parts = f" | {np.mean(t[:, 2] - t[:, 0]):5.2f} {np.mean(t[:, 3] - t[:, 2]):5.2f} {np.mean(t[:, 4] - t[:, 3]):5.2f} {np.mean(t[:, 1] - t[:, 4]):5.2f}"
        print(f"  {n:28s} {d.mean():7.2f}  ({d.min():6.2f} .. {d.max():6.2f}){parts}")
    per_layer = ts[1:, 0, 0] - ts[:-1, 0, 0]
    print(f"  layer to layer               {per_layer.mean():7.2f}  ({per_layer.min():6.2f} .. {per_layer.max():6.2f})")
    m.close()
