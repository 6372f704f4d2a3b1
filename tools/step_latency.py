#!/usr/bin/env python3
"""Per-token latency of the runner when a HOST is in the loop: one fused decode step per call, logits copied back and the next token
chosen on the host (the pattern of chatllm.cpp's generate loop), against the device-side greedy loop bench.py times (graphs back to
back, nothing returns to the host).  Shows how much of the drop-in path's per-token time is the single-step latency itself.
usage: python tools/step_latency.py [--steps 256] [--idle-us 0]   (--idle-us: host work simulated between tokens)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llama3-8b")
    ap.add_argument("--wtype", default="q4_k")
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--idle-us", type=float, default=0.0)
    a = ap.parse_args()
    pkg = bench.ge.load_package()
    cfg = pkg.synth.config(a.model, max_len=1024)
    m = bench.build_model(pkg, cfg, bench.WTYPES[a.wtype], 0, 1)
    prompt = np.arange(1, 17, dtype=np.int32)
    lg = m.forward(prompt)
    tok = int(np.argmax(lg))
    for _ in range(8):
        tok = int(np.argmax(m.decode_fused_logits(tok)))
    t0 = time.perf_counter()
    for _ in range(a.steps):
        lg = m.decode_fused_logits(tok)
        tok = int(np.argmax(lg))
        if a.idle_us:
            t1 = time.perf_counter()
            while (time.perf_counter() - t1) * 1e6 < a.idle_us:
                pass
    dt = (time.perf_counter() - t0) / a.steps
    print(f"one step per call (+{a.idle_us:.0f} us host idle): {dt*1e3:.3f} ms/token = {1/dt:.0f} tok/s")
    t0 = time.perf_counter()
    m.decode_greedy(tok, a.steps)
    dt = (time.perf_counter() - t0) / a.steps
    print(f"device-side greedy loop: {dt*1e3:.3f} ms/token = {1/dt:.0f} tok/s")


if __name__ == "__main__":
    main()
