#!/bin/bash
# round 5, call 12: attention + o projection as ONE launch (k_attn_o): parity tests, decode tok/s with CLLM_ATTN_O=0/1
O=gpurun_out/r5_12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv or attn or rope_kv" 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for a in 0 1 0 1; do
  CLLM_ATTN_O=$a timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attn_o=$a steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
for t in q4_0 q8_0; do for a in 0 1; do
  CLLM_ATTN_O=$a timeout 300 $B --steps 64 --warmup 8 --wtype $t 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t attn_o=$a steps64  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done; done
for a in 0 1; do
  CLLM_ATTN_O=$a timeout 600 $B --steps 32 --warmup 8 --model qwen2-72b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen2-72b attn_o=$a steps32  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
