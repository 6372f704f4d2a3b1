#!/bin/bash
# round 4, call 1: baseline, the Infinity-Cache-hot upper bound (every layer on layer 0's weights), the cross-launch touch prefetch by slice
O=gpurun_out/r4_1; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc"
run() { name=$1; shift; env "$@" $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      env "$@" $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-28s steps256 %.1f tok/s  tail %s  gate/up %.2f us' % ('$name', d['value'], d['greedy_tail'], d['roofline']['avg_us']))" | tee -a $O/summary.txt; }
run baseline X=1
run alias_layers CLLM_DEBUG_ALIAS_LAYERS=1
run touch31_sh7 CLLM_TOUCH=31
run touch31_sh6 CLLM_TOUCH=31 CLLM_TOUCH_SHIFT=6
for b in 1 2 4 8 16; do run touch$b CLLM_TOUCH=$b; done
run touch31_attn48 CLLM_TOUCH=31 CLLM_TOUCH_ATTN_MB=48
run touch31_attn16 CLLM_TOUCH=31 CLLM_TOUCH_ATTN_MB=16
run touch25 CLLM_TOUCH=25
run alias_touch31 CLLM_DEBUG_ALIAS_LAYERS=1 CLLM_TOUCH=31
