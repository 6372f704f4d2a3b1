#!/usr/bin/env python3
"""Which kernels does the decode step of a Qwen2-72B-shaped model launch, and how long do they take?  Two layers of the cfg4 shapes (Q4_K, down_proj Q8_0),
a few greedy tokens, launched eagerly (no graph) so that `rocprofv3 --kernel-trace --stats` sees every kernel by name.
usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q72 -- python /root/repo/tools/q72_kernel_mix.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

pkg = ge.load_package()
pkg.lib.require_gpu()
cfg = pkg.synth.config("qwen2-72b", max_len=256, n_layer=int(os.environ.get("LAYERS", "2")))
m = bench.build_model(pkg, cfg, pkg.Q4_K, 0, 1)
import numpy as np  # noqa: E402
m.use_graph(bool(int(os.environ.get("GRAPH", "0"))))
logits = m.forward(np.array([1, 5, 9, 200, 31, 7, 11, 300], np.int32), 0)
ids = m.decode_greedy(int(np.argmax(logits[-1])), int(os.environ.get("TOKENS", "24")), 8)
print("greedy ids", list(ids[-6:]))
import time  # noqa: E402
t0 = time.time()
n = int(os.environ.get("TOKENS", "24"))
ids = m.decode_greedy(int(ids[-1]), n)
dt = time.time() - t0
print(f"layers {cfg['n_layer']} graph {os.environ.get('GRAPH', '0')} team32 {os.environ.get('CLLM_GEMV_TEAM32', 'default')}: {dt / n * 1e6 / cfg['n_layer']:.1f} us per layer and token")
