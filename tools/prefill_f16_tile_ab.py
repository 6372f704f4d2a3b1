#!/usr/bin/env python3
"""cfg3 prefill (4096 tokens, Llama-3-8B shapes, Q4_0) in the f16 mode with dense_f16.hip's workgroup tile forced to 128 x 128 / 256 x 256 / picked, in ONE process
(box-to-box differences are larger than the effect)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
pkg.lib.require_gpu()
lib = pkg.lib.get()
dbg = C.CDLL(pkg.lib.SO_PATH)
n_prompt = 4096
cfg = pkg.synth.config("llama3-8b", max_len=n_prompt)
m = bench.build_model(pkg, cfg, bench.WTYPES["q4_0"], 0, 1)
prompt = np.random.default_rng(1234).integers(0, cfg["vocab"], n_prompt).astype(np.int32)
pkg.lib.check(lib.cllm_set_prefill_mode(0), "mode")
dbg.cllm_debug_set_prefill_f16(1)
for rnd in range(2):
    for tile in (128, 0, 256):
        dbg.cllm_debug_set_mmd_tile(tile)
        m.forward(prompt, n_past=0); pkg.ops.sync()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); m.forward(prompt, n_past=0); pkg.ops.sync(); ts.append(time.perf_counter() - t0)
        print(f"round {rnd} tile {tile or 'picked'}: median {sorted(ts)[1] * 1e3:.1f} ms", flush=True)
