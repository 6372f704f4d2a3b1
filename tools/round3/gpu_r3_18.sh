#!/bin/bash
O=gpurun_out/r3r; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "rope_kv_attn" 2>&1 | tail -5 | tee $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -3 | tee $O/pytest_llama.txt
for np in 600 2000 8000; do
  timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('two launches (soft_max inside V.P, V ring) n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
  CLLM_ATTN_LONG_3=1 timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('three launches n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
done
timeout 300 python bench.py --steps 512 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 (512 steps)', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
