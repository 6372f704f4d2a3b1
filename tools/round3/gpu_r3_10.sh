#!/bin/bash
O=gpurun_out/r3j; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "rope_kv_attn or rows32" 2>&1 | tail -6 | tee $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -4 | tee $O/pytest_llama.txt
for t in q4_0 q8_0 q4_1; do timeout 300 python bench.py --wtype $t --steps 128 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), 'tok/s')" | tee -a $O/decode_other_types.txt; done
for m in 1 0 2 4; do CLLM_GEMV_ROWS32=$m timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_0,q8_0 --iters 32 2>&1 | grep fused | sed "s/^/[q72 mode $m] /" | tee -a $O/gemv_fused_q72.txt; done
# long-context decode: attention launches' cost, 512 vs 1024 threshold
for thr in 512 1024 100000; do for np in 600 900 2000 8000; do CLLM_ATTN_LONG=$thr timeout 300 python bench.py --n-prompt $np --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('thr $thr n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/long_ctx_decode.txt; done; done
