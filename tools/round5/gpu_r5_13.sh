#!/bin/bash
# round 5, call 13: k_attn_o with arrival flags + gentle polling
O=gpurun_out/r5_13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for a in 0 1 0 1; do
  CLLM_ATTN_O=$a timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_o=$a steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
