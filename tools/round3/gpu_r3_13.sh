#!/bin/bash
O=gpurun_out/r3m; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "other_formats or get_rows or rope_kv_attn or rows32 or k_quants" 2>&1 | tail -6 | tee $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -q -x -k "long_context or fused_decode" 2>&1 | tail -3 | tee $O/pytest_llama.txt
timeout 300 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 --iters 64 2>&1 | grep fused | tee $O/gemv_fused_default.txt
for np in 600 2000 8000; do
  timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('exact(default) n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt
done
timeout 900 python bench.py --model qwen2-72b --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>$O/q72_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen2-72b shapes, runner:', round(d['value'],2), 'tok/s', d['config'])" | tee $O/cfg4.txt
tail -3 $O/q72_err.txt
timeout 1200 bash tools/dropin_qwen2_72b.sh 2>&1 | tee -a $O/cfg4.txt
