set -u
R=/root/repo; M=/tmp/mx_tiny.bin
python - <<'PY'
import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo/tests")
from conftest import load_package
gpu = load_package()
import make_ggmm
cfg = gpu.synth.config("tiny", max_len=64)
make_ggmm.write_mixtral("/tmp/mx_tiny.bin", cfg, 12, seed=91)
PY
cd $R/oracle/_ref
CLLM_HIP_SIG_DEBUG=1 CLLM_HIP_TRACE=1 ./ref_chat $M all 4 6 - 5 9 42 2> /tmp/mx_e.txt > /dev/null
grep "launch list differs" /tmp/mx_e.txt | tail -3
CLLM_HIP_STATS=1 ./ref_chat $M all 4 6 - 5 9 42 2> /tmp/mx_e2.txt > /dev/null
grep -E "graph_compute:|capture" /tmp/mx_e2.txt | cut -c1-200 | tail -6
