#!/usr/bin/env python3
"""stamps inside k_attn_long_scores during a real long-context decode (last layer, last step)"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402
pkg = ge.load_package(); L = pkg.lib.get(); pkg.lib.require_gpu()
lib = C.CDLL(pkg.lib.SO_PATH); lib.cllm_debug_set_attn_long_ts.argtypes = [C.c_void_p]
n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
cfg = pkg.synth.config("llama3-8b", max_len=(n_ctx + 64 + 63) // 64 * 64)
m = bench.build_model(pkg, cfg, 12, 0, 1)
ts = pkg.tensor.Buffer(512 * 8 * 8); L.cllm_memset(ts.ptr, 0, 512 * 64, None)
lib.cllm_debug_set_attn_long_ts(ts.ptr)
prompt = np.random.default_rng(1).integers(0, cfg["vocab"], n_ctx - 16).astype(np.int32)
tok = int(np.argmax(m.forward(prompt)))
m.decode_greedy(tok, 12)
host = np.zeros(512 * 8, dtype=np.uint64)
pkg.lib.check(L.cllm_memcpy_d2h(host.ctypes.data_as(C.c_void_p), ts.ptr, host.nbytes, None), "d2h"); L.cllm_stream_sync(None)
st = host.reshape(512, 8)[:256, :3].astype(np.int64); st = st[st[:, 2] > 0]
st = (st - st[:, 0].min()) / 100.0
for k, lab in enumerate(["entry", "rope + cache write done", "scores done"]):
    c = st[:, k]; print(f"  {k} {lab:26s} min {c.min():6.2f} median {np.median(c):6.2f} max {c.max():6.2f} us   ({len(c)} workgroups)")
