#!/bin/bash
O=gpurun_out/r3w; mkdir -p $O
for thr in 33 9 2; do
CLLM_MMF_EXACT_MIN_COLS=$thr python - <<'PY' 2>&1 | grep prompt | sed "s/^/[mmf_exact from $thr columns] /" | tee -a $O/ttft.txt
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
done
CLLM_MMF_EXACT_MIN_COLS=2 timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_ops.py -q -x -k "end_to_end or mul_mat_f or attention or long_prompt" 2>&1 | grep -E "passed|failed" | tee -a $O/ttft.txt
