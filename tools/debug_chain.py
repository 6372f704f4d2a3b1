"""dual chain: GPU ops on GPU-chain inputs vs oracle ops on oracle-chain inputs; prints where they separate"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle()
T, ops = pkg.Tensor, pkg.ops
def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
last = [0.0]
def cmp(name, g, r):
    e = rel(g, r)
    if e > 5 * max(last[0], 2e-7): print(f"    {name}: {last[0]:.1e} -> {e:.1e}", flush=True)
    last[0] = e
def olin(w, K, N, x):
    t, arr = w; q = x.shape[0]; want = np.zeros((q, N), np.float32)
    O.mul_mat(O.tensor(arr, t, [K, N]), O.tensor(np.ascontiguousarray(x), O.F32, [K, q]), O.tensor(want, O.F32, [N, q])); return want
def glin(w, K, N, x):
    t, arr = w; return ops.mul_mat(T.from_numpy(arr, t, [K, N]), T.from_numpy(x)).numpy().reshape(x.shape[0], N)
def onorm(wv, x, eps):
    q, H = x.shape; want = np.zeros_like(x); O.rms_norm(O.tensor(np.ascontiguousarray(x), O.F32, [H, q]), O.tensor(want, O.F32, [H, q]), eps); return want * wv
def gnorm(wv, x, eps): return ops.rms_norm_mul(T.from_numpy(x), T.from_numpy(wv), eps).numpy().reshape(x.shape)
def orope(arr, heads, hd, pos, cfg):
    q = arr.shape[0]; want = np.zeros_like(arr)
    O.rope(O.tensor(np.ascontiguousarray(arr), O.F32, [hd, heads, q]), pos, None, O.tensor(want, O.F32, [hd, heads, q]), hd, cfg["rope_mode"], cfg["rope_theta"]); return want
def grope(arr, heads, hd, pos, cfg):
    q = arr.shape[0]
    return ops.rope_ext(T.from_numpy(arr.reshape(q, heads, hd)), T.from_numpy(pos), None, hd, cfg["rope_mode"], 0, cfg["rope_theta"]).numpy().reshape(arr.shape)
def oattn(qq, kc, vc, cfg, q, n_past):
    hd, nh, nkv, ML = cfg["head_dim"], cfg["n_head"], cfg["n_kv_head"], cfg["max_len"]; KD = nkv*hd; n_kv = n_past + q
    sc = np.zeros((nh, q, n_kv), np.float32); ctx = np.zeros((nh, q, hd), np.float32); S = O.tensor(sc, O.F32, [n_kv, q, nh])
    O.mul_mat(O.tensor(kc, O.F16, [hd, n_kv, nkv], nb=[2, KD*2, hd*2, KD*ML*2]), O.tensor(qq, O.F32, [hd, q, nh], nb=[4, nh*hd*4, hd*4, nh*hd*q*4]), S)
    s0 = sc.copy()
    O.scale(S, S, 1.0/np.sqrt(hd)); O.diag_mask_inf(S, S, n_past); O.soft_max(S, None, S)
    O.mul_mat(O.tensor(vc, O.F16, [n_kv, hd, nkv], nb=[2, ML*2, ML*hd*2, ML*KD*2]), S, O.tensor(ctx, O.F32, [hd, q, nh]))
    return s0, sc, np.ascontiguousarray(ctx.transpose(1, 0, 2)).reshape(q, nh*hd)
def gattn(qq, kc, vc, cfg, q, n_past):
    hd, nh, nkv, ML = cfg["head_dim"], cfg["n_head"], cfg["n_kv_head"], cfg["max_len"]; KD = nkv*hd; n_kv = n_past + q
    dk, dv, dq = T.from_numpy(kc), T.from_numpy(vc), T.from_numpy(qq)
    s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD*2, hd*2]), dq.permute(0, 2, 1, 3)); s0 = s.numpy().reshape(nh, q, n_kv).copy()
    pg = ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past)
    c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), pg).numpy().reshape(nh, q, hd)
    return s0, pg.numpy().reshape(nh, q, n_kv), np.ascontiguousarray(c.transpose(1, 0, 2)).reshape(q, nh*hd)

def walk(cfg, w, seq):
    H, hd, nh, nkv, F, V, ML = cfg["hidden"], cfg["head_dim"], cfg["n_head"], cfg["n_kv_head"], cfg["ffn"], cfg["vocab"], cfg["max_len"]
    QD, KD = nh*hd, nkv*hd; NL = cfg["n_layer"]
    kcs = [np.zeros((NL, ML, KD), np.float16) for _ in range(2)]; vcs = [np.zeros((NL, KD, ML), np.float16) for _ in range(2)]
    n_past = 0
    for toks in seq:
        toks = np.asarray(toks, np.int32); q = toks.size; n_kv = n_past + q; pos = np.arange(n_past, n_kv, dtype=np.int32)
        print(f"  step n_past={n_past} qlen={q}", flush=True); last[0] = 0.0
        t, emb = w["tok_embd"]
        xg = ops.get_rows(T.from_numpy(emb, t, [H, V]), T.from_numpy(toks)).numpy().reshape(q, H); xr = xg.copy()
        for il in range(NL):
            p = f"layers.{il}."
            ng, nr = gnorm(w[p+"attn_norm"][1], xg, cfg["rms_eps"]), onorm(w[p+"attn_norm"][1], xr, cfg["rms_eps"]); cmp(f"L{il} attn_norm", ng, nr)
            qg, qr = glin(w[p+"wq"], H, QD, ng), olin(w[p+"wq"], H, QD, nr); cmp(f"L{il} wq", qg, qr)
            kg, kr = glin(w[p+"wk"], H, KD, ng), olin(w[p+"wk"], H, KD, nr); cmp(f"L{il} wk", kg, kr)
            vg, vr = glin(w[p+"wv"], H, KD, ng), olin(w[p+"wv"], H, KD, nr); cmp(f"L{il} wv", vg, vr)
            kg, kr = grope(kg, nkv, hd, pos, cfg), orope(kr, nkv, hd, pos, cfg); cmp(f"L{il} rope k", kg, kr)
            qg, qr = grope(qg, nh, hd, pos, cfg), orope(qr, nh, hd, pos, cfg); cmp(f"L{il} rope q", qg, qr)
            for (kc, vc, kk, vv) in ((kcs[0], vcs[0], kg, vg), (kcs[1], vcs[1], kr, vr)):
                kc[il, n_past:n_kv] = kk.astype(np.float16); vc[il][:, n_past:n_kv] = vv.T.astype(np.float16)
            cmp(f"L{il} kcache(f16)", kcs[0][il, :n_kv].astype(np.float32), kcs[1][il, :n_kv].astype(np.float32))
            sg, pg, ag = gattn(np.ascontiguousarray(qg.reshape(q, nh, hd)), kcs[0][il], vcs[0][il], cfg, q, n_past)
            sr, pr, ar = oattn(np.ascontiguousarray(qr.reshape(q, nh, hd)), kcs[1][il], vcs[1][il], cfg, q, n_past)
            cmp(f"L{il} scores", sg, sr); cmp(f"L{il} probs", pg, pr); cmp(f"L{il} attn out", ag, ar)
            og, orr = glin(w[p+"wo"], QD, H, ag), olin(w[p+"wo"], QD, H, ar); cmp(f"L{il} wo", og, orr)
            xg, xr = og + xg, orr + xr; cmp(f"L{il} resid1", xg, xr)
            ng, nr = gnorm(w[p+"ffn_norm"][1], xg, cfg["rms_eps"]), onorm(w[p+"ffn_norm"][1], xr, cfg["rms_eps"]); cmp(f"L{il} ffn_norm", ng, nr)
            gg, gr = glin(w[p+"wgate"], H, F, ng), olin(w[p+"wgate"], H, F, nr); cmp(f"L{il} wgate", gg, gr)
            ug, ur = glin(w[p+"wup"], H, F, ng), olin(w[p+"wup"], H, F, nr); cmp(f"L{il} wup", ug, ur)
            hg = ops.silu_mul(T.from_numpy(gg), T.from_numpy(ug)).numpy().reshape(q, F)
            hr = np.zeros_like(gr); O.silu(O.tensor(np.ascontiguousarray(gr), O.F32, [F, q]), O.tensor(hr, O.F32, [F, q])); hr = hr * ur; cmp(f"L{il} silu_mul", hg, hr)
            og, orr = glin(w[p+"wdown"], F, H, hg), olin(w[p+"wdown"], F, H, hr); cmp(f"L{il} wdown", og, orr)
            xg, xr = og + xg, orr + xr; cmp(f"L{il} resid2", xg, xr)
        ng, nr = gnorm(w["out_norm"][1], xg[-1:], cfg["rms_eps"]), onorm(w["out_norm"][1], xr[-1:], cfg["rms_eps"])
        lg, lr = glin(w["lm_head"], H, V, ng), olin(w["lm_head"], H, V, nr); cmp("lm_head", lg, lr)
        print(f"    logits max abs diff {np.max(np.abs(lg-lr)):.2e}", flush=True)
        n_past = n_kv

cfg = pkg.synth.config("tiny", max_len=64)
for wt, seed, plen in ((12, 1, 9), (8, 7, 1)):
    w = pkg.synth.make_model(cfg, wt, seed=seed)
    ref = O.Llama(cfg, w)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], plen).astype(np.int32)
    seq = [prompt]; lr = ref.forward(prompt)
    for s in range(4):
        t = int(np.argmax(lr)); seq.append([t]); lr = ref.forward([t])
    print("wt", wt, "seed", seed)
    walk(cfg, w, seq)
