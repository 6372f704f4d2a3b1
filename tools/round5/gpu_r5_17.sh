#!/bin/bash
# round 5, call 17: the sampler's first stage inside the lm_head launch + second stage / next embedding row / next cos-sin table as one launch (CLLM_DECODE_FOLD=0: the old step)
O=gpurun_out/r5_17; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_tp.py -m gpu -q -x 2>&1 | tail -4 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
run() { name=$1; fold=$2
      CLLM_DECODE_FOLD=$fold $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s steps20  %.1f tok/s  decode_512 %.1f  tail %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run old 0
run fold 1
run old 0
run fold 1
python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep "lm_head" | tee -a $O/summary.txt
