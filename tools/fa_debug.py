import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
gpu = load_package()
import make_ggmm
cfg = gpu.synth.config("small", max_len=128)
d = tempfile.mkdtemp(); mp = os.path.join(d, 'm.bin')
make_ggmm.write_model(mp, cfg, 12, seed=79)
env = dict(os.environ, REF_CHAT_FA="1", REF_CHAT_CACHE=sys.argv[1] if len(sys.argv) > 1 else "q8_0", GGML_SCHED_DEBUG="2", CLLM_HIP_STATS="1")
r = subprocess.run([os.path.join(ROOT, "oracle/_ref/ref_chat"), mp, "all", "4", "1", "-"] + [str(i) for i in range(3, 9)], capture_output=True, text=True, env=env)
print(r.returncode)
lines = (r.stderr + r.stdout).splitlines()
for ln in lines:
    if "SPLIT" in ln or "CPU" in ln or "graph_compute" in ln:
        print(ln[:260])
