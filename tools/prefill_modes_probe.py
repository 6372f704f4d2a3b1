"""Prefill numerics by mode (GPU): last-token logits of a 96-token prompt on the `small` model (4 layers, H 1024) against the oracle's CPU walk
(bit-identical to the reference host): default int8-MFMA path (exact integer block sums, own fp32 order), the opt-in dense fp16 path
(CLLM_PREFILL=f16), and the exact-order mat-vec path forced for every prompt length (CLLM_MMQ_MIN_COLS / CLLM_MMA_MIN_COLS = 100000).
usage: python tools/prefill_modes_probe.py          (spawns itself once per mode: the switches are read once per process)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def child(wt, out):
    from conftest import load_package
    gpu = load_package()
    gpu.lib.require_gpu()
    cfg = gpu.synth.config("small", max_len=128)
    m = gpu.Llama(cfg, gpu.synth.make_model(cfg, wt, seed=4))
    prompt = np.random.default_rng(4).integers(0, cfg["vocab"], 96).astype(np.int32)
    np.save(out, m.forward(prompt))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(int(sys.argv[1]), sys.argv[2])
        sys.exit(0)
    from conftest import load_package
    import oracle as O
    pkg = load_package()
    modes = (("int8 MFMA (default)", {}), ("dense fp16 (CLLM_PREFILL=f16)", {"CLLM_PREFILL": "f16"}),
             ("exact-order mat-vec for every length", {"CLLM_MMQ_MIN_COLS": "100000", "CLLM_MMA_MIN_COLS": "100000"}))
    for wt, name in ((O.Q4_0, "q4_0"), (O.Q4_K, "q4_k")):
        cfg = pkg.synth.config("small", max_len=128)
        ref = O.Llama(cfg, pkg.synth.make_model(cfg, wt, seed=4))
        prompt = np.random.default_rng(4).integers(0, cfg["vocab"], 96).astype(np.int32)
        want = ref.forward(prompt)
        for label, env in modes:
            out = f"/tmp/pm_{name}.npy"
            r = subprocess.run([sys.executable, __file__, str(wt), out], env=dict(os.environ, **env), capture_output=True, text=True)
            if r.returncode != 0:
                print(name, label, "FAILED", r.stderr[-300:])
                continue
            got = np.load(out)
            d = np.abs(got - want)
            print(f"{name} {label:40s}: max|dlogit| {d.max():.3e}  rms {np.sqrt(np.mean(d * d)):.3e}  (sigma of the logits {want.std():.3f})  argmax {'same' if got.argmax() == want.argmax() else 'DIFFERENT'}"
                  f"  words differing {int(np.sum(got.view(np.uint32) != want.view(np.uint32)))}/{want.size}", flush=True)
