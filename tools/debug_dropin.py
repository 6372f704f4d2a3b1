import sys, os, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as ge, make_ggmm
pkg = ge.load_package()
REF = os.path.join(ROOT, "oracle", "_ref", "ref_chat")
variants = [("tiny", {}), ("hd128", dict(head_dim=128, hidden=512)), ("gqa4", dict(n_head=8, n_kv_head=2, hidden=512)), ("H1024", dict(hidden=1024, n_head=16, n_kv_head=8)),
            ("F2816", dict(ffn=2816)), ("V2048", dict(vocab=2048)), ("L4", dict(n_layer=4)), ("ML512", dict(max_len=512))]
for name, over in variants:
    ml = over.pop("max_len", 64)
    cfg = pkg.synth.config("tiny", max_len=ml, **over)
    make_ggmm.write_model("/tmp/v.bin", cfg, 12, seed=5)
    prompt = [str(x) for x in (3, 100, 45, 260, 17, 9, 201, 5, 77, 12)]
    outs = {}
    for ngl in ("cpu", "all"):
        env = dict(os.environ)
        if ngl == "all": env["TEACHER"] = "/tmp/teacher.txt"
        r = subprocess.run([REF, "/tmp/v.bin", ngl, "4", "6", f"/tmp/l_{ngl}.bin"] + prompt, capture_output=True, text=True, env=env)
        if r.returncode: print(name, ngl, "FAILED", r.stderr[-500:]); break
        outs[ngl] = np.fromfile(f"/tmp/l_{ngl}.bin", np.float32).reshape(7, cfg["vocab"])
        if ngl == "cpu": open("/tmp/teacher.txt", "w").write(" ".join(r.stdout.split()))
    else:
        print(name, ["%.1e" % np.max(np.abs(outs["cpu"][i] - outs["all"][i])) for i in range(7)], flush=True)
