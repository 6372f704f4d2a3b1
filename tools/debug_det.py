import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from synth_helpers import rand_blocks
pkg = ge.load_package(); O = ge.load_oracle()
rng = np.random.default_rng(0)
# (a) GEMV determinism + correctness over many repeats
for t in (12, 8, 2):
    for K, N in ((256, 1024), (512, 256), (256, 320), (256, 768)):
        w = rand_blocks(t, N, K, rng); x = rng.standard_normal((1, K)).astype(np.float32)
        want = np.zeros((1, N), np.float32)
        O.mul_mat(O.tensor(w, t, [K, N]), O.tensor(x, O.F32, [K, 1]), O.tensor(want, O.F32, [N, 1]))
        dw, dx = pkg.Tensor.from_numpy(w, t, [K, N]), pkg.Tensor.from_numpy(x)
        first = None; nbad = 0; worst = 0
        for it in range(300):
            got = pkg.ops.mul_mat(dw, dx).numpy().reshape(1, N)
            if first is None: first = got.copy()
            if not np.array_equal(got, first): nbad += 1
            worst = max(worst, float(np.max(np.abs(got - want)) / np.max(np.abs(want))))
        print(f"gemv t={t} K={K} N={N}: nondeterministic repeats {nbad}/300 worst rel err {worst:.2e}", flush=True)
# (b) two model instances in lock step
cfg = pkg.synth.config("tiny", max_len=64)
for wt in (12, 8):
    w = pkg.synth.make_model(cfg, wt, seed=1)
    a, b = pkg.Llama(cfg, w), pkg.Llama(cfg, w)
    prompt = np.random.default_rng(1).integers(0, cfg["vocab"], 9).astype(np.int32)
    la, lb = a.forward(prompt), b.forward(prompt)
    res = []
    for s in range(30):
        res.append("=" if np.array_equal(la, lb) else f"{np.max(np.abs(la-lb)):.1e}")
        t = int(np.argmax(la))
        la, lb = a.forward([t]), b.forward([t])
    print("two instances wt", wt, " ".join(res), flush=True)
