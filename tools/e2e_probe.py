"""End-to-end exactness probe (GPU): whole models against the reference host's CPU run.
  1. the committed golden logits (tests/golden: chatllm.cpp's own CPU run of the tiny Llama-3 models): runner node path and fused decode path
  2. the drop-in module: oracle/_ref/ref_chat `cpu` vs `all` on synthetic GGMM files, FREE-RUNNING (no teacher forcing): ids and logits
Prints, per step, the number of differing logit words and max |delta|.  usage: python tools/e2e_probe.py [--big]"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from conftest import load_package  # noqa: E402
import oracle as O  # noqa: E402

gpu = load_package()
gpu.lib.get()
gpu.lib.require_gpu()
REF = os.path.join(ROOT, "oracle", "_ref")


def cmp(name, got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    out = []
    for s in range(want.shape[0]):
        bad = int(np.sum(got[s].view(np.uint32) != want[s].view(np.uint32)))
        out.append(f"{bad}/{float(np.max(np.abs(got[s] - want[s]))):.1e}")
    tot = int(np.sum(got.view(np.uint32) != want.view(np.uint32)))
    print(f"{'OK  ' if tot == 0 else 'DIFF'} {name}: {tot} words differ; per step (words/max|d|): {' '.join(out)}", flush=True)
    return tot


def golden():
    M = np.load(os.path.join(ROOT, "tests", "golden", "tiny_llama3_reference.npz"))
    G41 = np.load(os.path.join(ROOT, "tests", "golden", "q4_1_reference.npz"))
    tot = 0
    for t, name in ((O.Q4_K, "q4_k"), (O.Q4_0, "q4_0"), (O.Q8_0, "q8_0"), (O.Q4_1, "q4_1")):
        prompt, ids, logits = (G41["model_prompt"], G41["model_ids"], G41["model_logits"]) if name == "q4_1" else (M["prompt"], M[f"{name}_ids"], M[f"{name}_logits"])
        cfg = gpu.synth.config("tiny", max_len=64)
        for path in ("nodes", "fused"):
            m = gpu.Llama(cfg, gpu.synth.make_model(cfg, t, seed=1234))
            got = [m.forward(prompt)]
            for s in range(12):
                got.append(m.forward([int(ids[s])]) if path == "nodes" else m.decode_fused_logits(int(ids[s])))
            tot += cmp(f"golden {name} runner/{path}", np.stack(got), logits[:13])
            m.close()
    return tot


def dropin(arch, cfgname, wt, n_dec, prompt, over=None, threads=8, extra_env=None, tag=""):
    import make_ggmm
    cfg = gpu.synth.config(cfgname, **(over or {}))
    d = tempfile.mkdtemp(prefix="e2e_")
    mp = os.path.join(d, "m.bin")
    make_ggmm.write_model(mp, cfg, wt, seed=77, arch=arch, fast=cfgname not in ("tiny", "small", "gpt2s-llama"))

    def run(ngl):
        lp = os.path.join(d, f"l_{ngl}.bin")
        env = dict(os.environ, **(extra_env or {}))
        r = subprocess.run([os.path.join(REF, "ref_chat"), mp, ngl, str(threads), str(n_dec), lp] + [str(p) for p in prompt], capture_output=True, text=True, env=env, timeout=3000)
        if r.returncode != 0:
            print("ref_chat failed:", r.stderr[-1500:])
            raise SystemExit(1)
        return [int(x) for x in r.stdout.split()], np.fromfile(lp, np.float32).reshape(n_dec + 1, cfg["vocab"]), r.stderr

    ids_c, lg_c, _ = run("cpu")
    ids_g, lg_g, err = run("all")
    mism = sum(int(a != b) for a, b in zip(ids_c, ids_g))
    within = float(np.mean(np.max(np.abs(lg_c - lg_g), axis=1) <= 1e-3))
    tot = cmp(f"dropin {arch}/{cfgname}/{wt}{tag} free-running: greedy id mismatches {mism}/{len(ids_c)}, steps within 1e-3: {within:.3f}", lg_g, lg_c)
    for ln in err.splitlines():
        if "decode:" in ln:
            print("     ", ln.strip())
    os.remove(mp)
    return tot


if __name__ == "__main__":
    total = golden()
    for wt in (12, 2, 8, 3):
        total += dropin("llama3", "tiny", wt, 24, [3, 100, 45, 260, 17, 9, 201], over=dict(max_len=64))
    total += dropin("qwen2", "tiny", 12, 24, [3, 100, 45, 260, 17, 9, 201], over=dict(max_len=64, qkv_bias=1, rope_mode=2, rope_theta=1e6))
    total += dropin("llama3", "gpt2s-llama", 8, 64, list(range(5, 21)))
    if "--big" in sys.argv:
        total += dropin("llama3", "llama3-8b", 12, 64, [(7 * i + 3) % 128000 for i in range(16)], over=dict(max_len=1024), threads=32, tag=" (BASELINE cfg2 shapes)")
    print("TOTAL words differing:", total)
