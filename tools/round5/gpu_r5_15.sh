#!/bin/bash
# round 5, call 14: k_attn_o with the whole o projection prefetched (two steps per wave) behind a start delay; delay sweep
O=gpurun_out/r5_15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
CLLM_ATTN_O=0 timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_o=0 steps20  %.1f tok/s  decode_512 %.1f' % (d['value'], d['decode_512']['value']))" | tee -a $O/summary.txt
for dl in 0 5 5; do
  CLLM_ATTN_O=1 CLLM_ATTN_O_DELAY=$dl timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_o=1 delay=$dl steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
CLLM_ATTN_O=0 timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_o=0 steps20  %.1f tok/s  decode_512 %.1f' % (d['value'], d['decode_512']['value']))" | tee -a $O/summary.txt
