#!/bin/bash
# round 6, call 20: what the reference's accumulation ORDER costs the headline (Q4_K) decode: the same kernels with the free fp32 fold (CLLM_DECODE_FREE_ORDER=2: a pricing
# experiment, never a product path -- the logits are far off), step rate and the per-launch table, against the default
O=gpurun_out/r6_20; mkdir -p $O
for fo in 0 2; do
  CLLM_DECODE_FREE_ORDER=$fo timeout 600 python bench.py --steps 20 --warmup 5 --no-pmc --no-prefill --no-cpu-baseline --no-other-types 2>$O/err_$fo.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CLLM_DECODE_FREE_ORDER=$fo: %.1f tok/s  %.4f ms/step  tail %s' % (d['value'], d['ms_per_step'], d.get('greedy_tail')))
for k in d.get('kernels', {}).get('per_launch', []): print('   %-80s %7.2f us  %6.1f GB/s  %.3f' % (k['name'][:80], k['us'], k['gbs'], k['frac']))" | tee -a $O/price.txt
done
