import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
over = dict(rope_mode=2, qkv_bias=1, rope_theta=1e6, ffn=544, n_layer=1)
cfg = pkg.synth.config("tiny", max_len=64, **over)
H, hd, nh, nkv, F = cfg["hidden"], cfg["head_dim"], cfg["n_head"], cfg["n_kv_head"], cfg["ffn"]
QKV = (nh + 2*nkv) * hd
found = 0
for seed in range(1, 30):
    w = pkg.synth.make_model(cfg, 12, seed=seed)
    a, b = pkg.Llama(cfg, w), pkg.Llama(cfg, w)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], 9).astype(np.int32)
    a.forward(prompt); b.forward(prompt)
    for step, t in enumerate(np.random.default_rng(seed + 100).integers(0, cfg["vocab"], 50)):
        la, lb = a.forward([int(t)]), b.decode_fused_logits(int(t))
        bufs = {}
        for what, n in (("qkv", QKV), ("att", nh*hd), ("gu", 2*F), ("x", H)):
            ba, bb = a.debug_read(what, n), b.debug_read(what, n)
            d = np.flatnonzero(ba != bb)
            if len(d): bufs[what] = (len(d), d[:6].tolist(), ba[d[:3]].tolist(), bb[d[:3]].tolist())
        if bufs or not np.array_equal(la, lb):
            print("seed", seed, "step", step, "n_past", a.n_past - 1, "logits equal", np.array_equal(la, lb), bufs, flush=True)
            found += 1
            break
    a.close(); b.close()
    if found >= 4: break
print("done, found", found)
