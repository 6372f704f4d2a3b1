"""op-by-op walk of the decoder on the GPU ops with the oracle evaluated on the SAME inputs after every op"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
pkg = ge.load_package(); O = ge.load_oracle()
T, ops = pkg.Tensor, pkg.ops

def rel(a, b): return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))
def report(name, g, r, thr=1e-5):
    e = rel(g, r)
    if e > thr: print(f"    !! {name}: rel err {e:.2e}", flush=True)
    return e

CH = {}
def chain(name, g, r_same_input, xr_out):
    """g: gpu output (gpu chain), xr_out: oracle output on the ORACLE chain's input"""
    e = rel(g, xr_out)
    prev = CH.get("last", 0.0)
    if e > 10 * max(prev, 1e-7): print(f"    chain divergence at {name}: {prev:.1e} -> {e:.1e}", flush=True)
    CH["last"] = e

def olinear(w, K, N, x):
    t, arr = w; q = x.shape[0]
    want = np.zeros((q, N), np.float32)
    O.mul_mat(O.tensor(arr, t, [K, N]), O.tensor(np.ascontiguousarray(x), O.F32, [K, q]), O.tensor(want, O.F32, [N, q]))
    return want

def onorm(wv, x, eps):
    q, H = x.shape; want = np.zeros_like(x)
    O.rms_norm(O.tensor(np.ascontiguousarray(x), O.F32, [H, q]), O.tensor(want, O.F32, [H, q]), eps)
    return want * wv

def linear(w, K, N, x):           # x [qlen, K] numpy
    t, arr = w
    q = x.shape[0]
    want = np.zeros((q, N), np.float32)
    O.mul_mat(O.tensor(arr, t, [K, N]), O.tensor(np.ascontiguousarray(x), O.F32, [K, q]), O.tensor(want, O.F32, [N, q]))
    got = ops.mul_mat(T.from_numpy(arr, t, [K, N]), T.from_numpy(x)).numpy().reshape(q, N)
    return got, want

def norm(wv, x, eps):
    q, H = x.shape
    want = np.zeros_like(x)
    O.rms_norm(O.tensor(np.ascontiguousarray(x), O.F32, [H, q]), O.tensor(want, O.F32, [H, q]), eps)
    want = want * wv
    got = ops.rms_norm_mul(T.from_numpy(x), T.from_numpy(wv), eps).numpy().reshape(q, H)
    return got, want

def walk(cfg, w, tokens_list):
    H, hd, nh, nkv, F, V, ML = cfg["hidden"], cfg["head_dim"], cfg["n_head"], cfg["n_kv_head"], cfg["ffn"], cfg["vocab"], cfg["max_len"]
    QD, KD = nh*hd, nkv*hd
    kc = np.zeros((cfg["n_layer"], ML, KD), np.float16); vc = np.zeros((cfg["n_layer"], KD, ML), np.float16)
    n_past = 0
    for toks in tokens_list:
        toks = np.asarray(toks, np.int32); q = toks.size; n_kv = n_past + q
        print(f"  step n_past={n_past} qlen={q}", flush=True)
        t, emb = w["tok_embd"]
        x = ops.get_rows(T.from_numpy(emb, t, [H, V]), T.from_numpy(toks)).numpy().reshape(q, H)
        pos = np.arange(n_past, n_kv, dtype=np.int32)
        for il in range(cfg["n_layer"]):
            p = f"layers.{il}."
            xn, r = norm(w[p+"attn_norm"][1], x, cfg["rms_eps"]); report(f"L{il} attn_norm", xn, r, 1e-6)
            qv, r = linear(w[p+"wq"], H, QD, xn); report(f"L{il} wq", qv, r)
            kv, r = linear(w[p+"wk"], H, KD, xn); report(f"L{il} wk", kv, r)
            vv, r = linear(w[p+"wv"], H, KD, xn); report(f"L{il} wv", vv, r)
            for nm, arr, heads in (("k", kv, nkv), ("q", qv, nh)):
                want = np.zeros_like(arr)
                O.rope(O.tensor(np.ascontiguousarray(arr), O.F32, [hd, heads, q]), pos, None, O.tensor(want, O.F32, [hd, heads, q]), hd, cfg["rope_mode"], cfg["rope_theta"])
                got = ops.rope_ext(T.from_numpy(arr.reshape(q, heads, hd)), T.from_numpy(pos), None, hd, cfg["rope_mode"], 0, cfg["rope_theta"]).numpy().reshape(arr.shape)
                report(f"L{il} rope {nm}", got, want, 1e-5)
                if nm == "k": kv = got
                else: qv = got
            kc[il, n_past:n_kv] = kv.astype(np.float16); vc[il][:, n_past:n_kv] = vv.T.astype(np.float16)
            # attention
            qq = np.ascontiguousarray(qv.reshape(q, nh, hd))
            sc = np.zeros((nh, q, n_kv), np.float32); ctx = np.zeros((nh, q, hd), np.float32)
            S = O.tensor(sc, O.F32, [n_kv, q, nh])
            O.mul_mat(O.tensor(kc[il], O.F16, [hd, n_kv, nkv], nb=[2, KD*2, hd*2, KD*ML*2]), O.tensor(qq, O.F32, [hd, q, nh], nb=[4, nh*hd*4, hd*4, nh*hd*q*4]), S)
            dk, dv, dq = T.from_numpy(kc[il]), T.from_numpy(vc[il]), T.from_numpy(qq)
            s = ops.mul_mat(dk.view([hd, n_kv, nkv], [2, KD*2, hd*2]), dq.permute(0, 2, 1, 3))
            report(f"L{il} scores", s.numpy().reshape(sc.shape), sc)
            sg = s.numpy().reshape(sc.shape).copy()
            pr = sg.copy(); Pm = O.tensor(pr, O.F32, [n_kv, q, nh])
            O.scale(Pm, Pm, 1.0/np.sqrt(hd)); O.diag_mask_inf(Pm, Pm, n_past); O.soft_max(Pm, None, Pm)
            pg = ops.scale_mask_soft_max(s, 1.0/np.sqrt(hd), n_past)
            report(f"L{il} probs", pg.numpy().reshape(pr.shape), pr, 1e-6)
            pgn = pg.numpy().reshape(pr.shape).copy()
            O.mul_mat(O.tensor(vc[il], O.F16, [n_kv, hd, nkv], nb=[2, ML*2, ML*hd*2, ML*KD*2]), O.tensor(pgn, O.F32, [n_kv, q, nh]), O.tensor(ctx, O.F32, [hd, q, nh]))
            c = ops.mul_mat(dv.view([n_kv, hd, nkv], [2, ML*2, ML*hd*2]), pg)
            cg = c.numpy().reshape(ctx.shape); report(f"L{il} ctx", cg, ctx)
            att = np.ascontiguousarray(cg.transpose(1, 0, 2)).reshape(q, QD)
            o, r = linear(w[p+"wo"], QD, H, att); report(f"L{il} wo", o, r)
            x = o + x
            xn, r = norm(w[p+"ffn_norm"][1], x, cfg["rms_eps"]); report(f"L{il} ffn_norm", xn, r, 1e-6)
            g, r = linear(w[p+"wgate"], H, F, xn); report(f"L{il} wgate", g, r)
            u, r = linear(w[p+"wup"], H, F, xn); report(f"L{il} wup", u, r)
            want = np.zeros_like(g); O.silu(O.tensor(np.ascontiguousarray(g), O.F32, [F, q]), O.tensor(want, O.F32, [F, q])); want = want * u
            gg = ops.silu_mul(T.from_numpy(g), T.from_numpy(u)).numpy().reshape(q, F); report(f"L{il} silu_mul", gg, want, 1e-6)
            o, r = linear(w[p+"wdown"], F, H, gg); report(f"L{il} wdown", o, r)
            x = o + x
        xn, r = norm(w["out_norm"][1], x[-1:], cfg["rms_eps"])
        lg, r = linear(w["lm_head"], H, V, xn); report("lm_head", lg, r)
        n_past = n_kv
        yield lg[0]

cfg = pkg.synth.config("tiny", max_len=64)
for wt, seed in ((12, 1), (8, 7)):
    w = pkg.synth.make_model(cfg, wt, seed=seed)
    ref = O.Llama(cfg, w); dev = pkg.Llama(cfg, w)
    prompt = np.random.default_rng(seed).integers(0, cfg["vocab"], 9 if wt == 12 else 1).astype(np.int32)
    seq = [prompt]; lr = ref.forward(prompt); ld = dev.forward(prompt)
    for s in range(5):
        t = int(np.argmax(lr)); seq.append([t]); lr = ref.forward([t]); ld = dev.forward([t])
    print("wt", wt, "seed", seed, "decoder-vs-oracle final", f"{np.max(np.abs(lr-ld)):.2e}")
    lw = None
    for lw in walk(cfg, w, seq): pass
    print("  walk-vs-oracle final", f"{np.max(np.abs(lw-lr)):.2e}", "walk-vs-decoder", f"{np.max(np.abs(lw-ld)):.2e}")
