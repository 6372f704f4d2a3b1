#!/bin/bash
O=gpurun_out/r3v; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -22 | tee $O/pytest_gpu_full.txt
python - <<'PY' 2>&1 | tee $O/ttft.txt
import sys, time, numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as ge, bench
pkg = ge.load_package(); pkg.lib.require_gpu()
cfg = pkg.synth.config("llama3-8b", max_len=1024)
m = bench.build_model(pkg, cfg, bench.WTYPES["q4_k"], 0, 1)
for n in (4, 8, 12, 16, 24, 32, 64):
    prompt = np.random.default_rng(n).integers(0, cfg["vocab"], n).astype(np.int32)
    m.forward(prompt, n_past=0); pkg.ops.sync()
    t0 = time.perf_counter()
    for _ in range(3): m.forward(prompt, n_past=0)
    pkg.ops.sync()
    print(f"prompt of {n:3d} tokens (8B shapes, Q4_K, exact): {(time.perf_counter()-t0)/3*1e3:7.2f} ms")
PY
