import os, subprocess, sys, tempfile
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, ROOT+"/tools"); sys.path.insert(0, ROOT+"/tests")
from conftest import load_package
gpu = load_package()
import make_ggmm
cfg = gpu.synth.config("tiny", max_len=64)
d = tempfile.mkdtemp(); mp = os.path.join(d, 'm.bin')
make_ggmm.write_model(mp, cfg, 8, seed=77)
def run(ngl, **env):
    lp = os.path.join(d, "l.bin")
    r = subprocess.run([ROOT+"/oracle/_ref/ref_chat", mp, ngl, "4", "12", lp, "3","100","45","260","17","9","201"], capture_output=True, text=True, env=dict(os.environ, **env))
    return [int(x) for x in r.stdout.split()], np.fromfile(lp, np.float32).reshape(13, cfg["vocab"]), r.stderr
ic, lc, _ = run("cpu")
for name, env in (("ahead off", {"CLLM_HIP_AHEAD":"0"}), ("ahead on", {}), ("ahead sync", {"CLLM_HIP_AHEAD_SYNC":"1"})):
    ig, lg, err = run("all", CLLM_HIP_STATS="1", **env)
    print(name, [int(np.sum(lg[s].view(np.uint32) != lc[s].view(np.uint32))) for s in range(13)], ig == ic)
ig, lg, err = run("all", CLLM_HIP_AHEAD_SYNC="1")
for s in range(3, 8):
    bad = np.nonzero(lg[s].view(np.uint32) != lc[s].view(np.uint32))[0]
    print("step", s, "token fed", ig[s-1] if s > 0 else None, "next", ig[s], "bad idx", bad[:5], "gpu", lg[s][bad[:3]], "cpu", lc[s][bad[:3]], "argmax cpu", int(np.argmax(lc[s])))
