#!/bin/bash
# round 5, call 34: where the one-launch attention (k_attn_dec, now cheaper) hands over to the split attention (attn_long.hip): decode at 600 / 800 / 1000 cached positions with the threshold at 512 (default) and 1024
O=gpurun_out/r5_34; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --no-prefill --no-kernels --steps 24 --warmup 8"
for np in 580 780 980; do
  for thr in 512 1024; do
    CLLM_ATTN_LONG=$thr $B --n-prompt $np 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('n_prompt $np threshold $thr  %.1f tok/s  n_ctx_end %s tail %s' % (d['value'], d['config'].get('n_ctx_end'), d['greedy_tail']))" | tee -a $O/summary.txt
  done
done
