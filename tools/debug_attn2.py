import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
T, ops = pkg.Tensor, pkg.ops
rng = np.random.default_rng(0)
for (hd, nh, nkv, ML) in [(64, 4, 2, 64), (128, 8, 2, 512)]:
    KD = hd * nkv
    bad = 0
    for trial in range(400):
        n_past = int(rng.integers(0, ML - 1)); n_kv = n_past + 1
        kc = rng.standard_normal((ML, KD)).astype(np.float16); vc = rng.standard_normal((KD, ML)).astype(np.float16)
        q = rng.standard_normal((1, nh, hd)).astype(np.float32)
        dk, dv, dq = T.from_numpy(kc), T.from_numpy(vc), T.from_numpy(q)
        s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD*2, hd*2]), dq.permute(0, 2, 1, 3))
        p = ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past, dst=s)
        c = ops.cont(ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), p).permute(0, 2, 1, 3)).numpy().reshape(-1)
        f = ops.attn_decode(dq, T.from_numpy(np.array([n_past], np.int32)), nh, nkv, hd, dk, dv, ML).numpy().reshape(-1)
        if not np.array_equal(c, f):
            bad += 1
            d = np.flatnonzero(c != f)
            if bad <= 5: print(f"hd={hd} n_kv={n_kv}: {len(d)} differ, heads {sorted(set((d // hd).tolist()))}, max abs {np.max(np.abs(c-f)):.2e}", flush=True)
    print(f"hd={hd}: {bad}/400 trials differ", flush=True)
