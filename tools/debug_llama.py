import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle()
def run(name, wt, plen, nsteps, seed, ml=64, **over):
    cfg = pkg.synth.config(name, max_len=ml, **over)
    w = pkg.synth.make_model(cfg, wt, seed=seed)
    ref, dev = O.Llama(cfg, w), pkg.Llama(cfg, w)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], plen).astype(np.int32)
    lr, lg = ref.forward(prompt), dev.forward(prompt)
    out = []
    for s in range(nsteps):
        d = lr - lg
        out.append(f"{np.max(np.abs(d)):.1e}/{np.linalg.norm(d)/np.linalg.norm(lr):.1e}")
        t = int(np.argmax(lr))
        lr, lg = ref.forward([t]), dev.forward([t])
    print(name, wt, "plen", plen, "seed", seed, " ".join(out), flush=True)
    dev.close()
for seed in (1, 2, 3):
    for wt in (8, 12, 2):
        run("tiny", wt, 9, 20, seed)
run("tiny", 8, 1, 24, 7)
run("tiny", 12, 3, 24, 8)
run("small", 12, 8, 4, 1, ml=96)
run("small", 12, 9, 4, 1, ml=96)
run("small", 12, 40, 4, 1, ml=96)
os.environ["CLLM_NO_MMQ"] = "1"
run("small", 12, 40, 4, 1, ml=96)
run("tiny", 12, 9, 20, 1)
