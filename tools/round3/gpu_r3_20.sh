#!/bin/bash
O=gpurun_out/r3t; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "rope_kv_attn" 2>&1 | tail -3 | tee $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_llama.py tests/test_gpu_tp.py -q -x 2>&1 | tail -3 | tee $O/pytest_llama_tp.txt
for np in 600 2000 8000; do
  timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact (two launches, 16-wave soft_max) n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_l; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o p -- python /root/repo/bench.py --n-prompt 8000 --steps 32 --warmup 4 --no-cpu-baseline --no-pmc --no-graph > /root/repo/$O/long_8000.log 2>&1
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|attn_long" "$f" > /root/repo/$O/long_8000_kernel_stats.csv
python3 - <<'PY'
import csv
for r in csv.reader(open("/root/repo/gpurun_out/r3t/long_8000_kernel_stats.csv")):
    if r[0] != "Name": print(r[0].split("(")[0][:50], r[1], round(float(r[3])/1e3, 2), "us")
PY
