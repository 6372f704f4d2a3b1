#!/bin/bash
O=gpurun_out/r3aa; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py -q -x -k "rope_kv_attn or attn_decode or fused_decode or end_to_end or persistent" 2>&1 | tail -3 | tee $O/pytest.txt
for i in 1 2; do timeout 300 python bench.py --steps 512 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg2 (512 steps)', round(d['value'],1), 'tok/s')" | tee -a $O/bench.txt; done
timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('128 steps', round(d['value'],1), 'tok/s')" | tee -a $O/bench.txt
