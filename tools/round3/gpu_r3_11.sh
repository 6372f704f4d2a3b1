#!/bin/bash
O=gpurun_out/r3k; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "rope_kv_attn" 2>&1 | tail -6 | tee $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -4 | tee $O/pytest_llama.txt
for np in 600 900 2000 8000; do
  timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact(default) n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
  CLLM_ATTN_LONG_FLASH=1 timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('flash(opt-in)  n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
done
CLLM_ATTN_LONG=1024 timeout 300 python bench.py --n-prompt 900 --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one-launch (threshold 1024) n_prompt 900', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
