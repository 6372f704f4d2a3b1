import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle()
rng = np.random.default_rng(3)
def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
for (hd, nh, nkv, ML) in [(64, 4, 2, 64), (128, 8, 2, 96)]:
    KD = hd * nkv
    kc = rng.standard_normal((ML, KD)).astype(np.float16)
    vc = rng.standard_normal((KD, ML)).astype(np.float16)
    dk, dv = pkg.Tensor.from_numpy(kc), pkg.Tensor.from_numpy(vc)
    for qlen, n_past in [(1, p) for p in range(0, 40)] + [(40, 0), (9, 0)]:
        n_kv = n_past + qlen
        q = rng.standard_normal((qlen, nh, hd)).astype(np.float32)
        sc = np.zeros((nh, qlen, n_kv), np.float32); pr = np.zeros_like(sc); ctx = np.zeros((nh, qlen, hd), np.float32)
        S = O.tensor(sc, O.F32, [n_kv, qlen, nh]); Pm = O.tensor(pr, O.F32, [n_kv, qlen, nh])
        O.mul_mat(O.tensor(kc, O.F16, [hd, n_kv, nkv], nb=[2, KD*2, hd*2, KD*ML*2]), O.tensor(q, O.F32, [hd, qlen, nh], nb=[4, nh*hd*4, hd*4, nh*hd*qlen*4]), S)
        O.scale(S, Pm, 1.0/np.sqrt(hd)); O.diag_mask_inf(Pm, Pm, n_past); O.soft_max(Pm, None, Pm)
        O.mul_mat(O.tensor(vc, O.F16, [n_kv, hd, nkv], nb=[2, ML*2, ML*hd*2, ML*KD*2]), Pm, O.tensor(ctx, O.F32, [hd, qlen, nh]))
        dq = pkg.Tensor.from_numpy(q)
        s = pkg.ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD*2, hd*2]), dq.permute(0, 2, 1, 3))
        e1 = rel(s.numpy().reshape(sc.shape), sc)
        p = pkg.ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past)
        e2 = rel(p.numpy().reshape(pr.shape), pr)
        pin = pkg.ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past, dst=s)    # in place like the decoder
        e2b = rel(pin.numpy().reshape(pr.shape), pr)
        c = pkg.ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), p)
        e3 = rel(c.numpy().reshape(ctx.shape), ctx)
        # V.P against the oracle's own probabilities (isolates the second matmul)
        c2 = pkg.ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), pkg.Tensor.from_numpy(pr))
        e4 = rel(c2.numpy().reshape(ctx.shape), ctx)
        flag = " <<<<" if max(e1, e2, e2b, e3, e4) > 1e-4 else ""
        print(f"hd={hd} qlen={qlen} n_past={n_past} n_kv={n_kv}: scores {e1:.2e} probs {e2:.2e} inplace {e2b:.2e} ctx {e3:.2e} ctx(oracle P) {e4:.2e}{flag}")
