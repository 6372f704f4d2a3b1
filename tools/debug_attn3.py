import sys, os, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package()
T, ops = pkg.Tensor, pkg.ops
L = pkg.lib.get()
rng = np.random.default_rng(0)
hd, nh, nkv, ML = 64, 4, 2, 64
KD = hd * nkv
dbg = T(pkg.F32, [nh * ML + 2 * nh])
L.cllm_debug_set_attn_probs.argtypes = [C.c_void_p]; L.cllm_debug_set_attn_probs(dbg.data_ptr())
shown = 0
for trial in range(400):
    n_past = int(rng.integers(0, ML - 1)); n_kv = n_past + 1
    kc = rng.standard_normal((ML, KD)).astype(np.float16); vc = rng.standard_normal((KD, ML)).astype(np.float16)
    q = rng.standard_normal((1, nh, hd)).astype(np.float32)
    dk, dv, dq = T.from_numpy(kc), T.from_numpy(vc), T.from_numpy(q)
    s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD*2, hd*2]), dq.permute(0, 2, 1, 3))
    sraw = s.numpy().reshape(nh, n_kv).copy()
    p = ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past)
    pn = p.numpy().reshape(nh, n_kv)
    c = ops.cont(ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), p).permute(0, 2, 1, 3)).numpy().reshape(-1)
    f = ops.attn_decode(dq, T.from_numpy(np.array([n_past], np.int32)), nh, nkv, hd, dk, dv, ML).numpy().reshape(-1)
    d = dbg.numpy().reshape(-1)
    pf = d[:nh*ML].reshape(nh, ML)[:, :n_kv]; extra = d[nh*ML:].reshape(nh, 2)
    if not np.array_equal(c, f) and shown < 3:
        shown += 1
        h = int(np.flatnonzero(c != f)[0] // hd)
        dp = np.flatnonzero(pn[h].astype(np.float16).astype(np.float32) != pf[h])
        # emulate V.P in the documented order (G=8 for n_kv < 128): lane gl: tail element then its 8-chunk; xor butterfly 4,2,1
        p16 = pn[h].astype(np.float16).astype(np.float32)
        g = h // (nh // nkv); emu = np.zeros(hd, np.float32)
        n8 = n_kv & ~7
        for dd in range(hd):
            vr = vc[g*hd + dd].astype(np.float32)
            acc = np.zeros(8, np.float32)
            for gl in range(8):
                a = np.float32(0)
                i = n8 + gl
                while i < n_kv: a = np.float32(a + vr[i]*p16[i]); i += 8
                i0 = gl*8
                while i0 < n8:
                    for j in range(8): a = np.float32(a + np.float32(vr[i0+j]*p16[i0+j]))
                    i0 += 64
                acc[gl] = a
            for o in (4, 2, 1): acc = np.array([np.float32(acc[l] + acc[l ^ o]) for l in range(8)], np.float32)
            emu[dd] = acc[0]
        cg, fg = c[h*hd:(h+1)*hd], f[h*hd:(h+1)*hd]
        print("   emu==general", np.array_equal(emu, cg), "emu==fused", np.array_equal(emu, fg), "n differ g/f", int((emu != cg).sum()), int((emu != fg).sum()))
        print(f"n_kv={n_kv} head {h}: probs differ at {len(dp)} positions {dp[:8]}; general {pn[h][dp[:3]]} fused {pf[h][dp[:3]]}; mx {extra[h,0]} vs {np.float32(sraw[h].max()*np.float32(1.0/np.sqrt(hd)))} inv {extra[h,1]} sum(pn) {pn[h].sum()}", flush=True)
