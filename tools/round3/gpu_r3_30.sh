#!/bin/bash
O=gpurun_out/r3ad; mkdir -p $O
for np in 150 250 350 450; do
  for thr in 64 100000; do
    CLLM_ATTN_LONG=$thr timeout 300 python bench.py --n-prompt $np --steps 32 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threshold $thr n_prompt $np', round(d['value'],1), 'tok/s')" | tee -a $O/attn_crossover.txt
  done
done
