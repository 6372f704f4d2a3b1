#!/bin/bash
# round 6, call 19: the opt-in free-order tier of Q4_0 / Q4_1 / Q8_0 decode: small-shape tests, the 8B-shape comparison against the exact order, tok/s of both
O=gpurun_out/r6_19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py -x -q -m gpu -k "free_order" -s 2>&1 | tail -15 | tee $O/pytest_small.txt
timeout 1500 python -m pytest tests/test_gpu_dropin.py -q -m gpu -k "free_order_tier_at" -s 2>&1 | tail -40 | grep -v "^E\|^ " | tee $O/pytest_8b.txt
timeout 900 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill 2>$O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('q4_k', round(d['value'], 1))
for k, v in d['other_types'].items(): print(k, round(v.get('value', 0), 1), round(v.get('model_hbm_frac', 0), 3), v.get('greedy_tail'))" | tee $O/other_types.txt
