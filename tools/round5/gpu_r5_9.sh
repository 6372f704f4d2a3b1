#!/bin/bash
# round 5, call 9: the one-shot all-reduce on data-tagged granules + the all-reduce fused into the mat-vecs (two processes on the GPU), no change on the single-GPU kernels
O=gpurun_out/r5_9; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_tp.py -m gpu -q -x 2>&1 | tail -15 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv or llama or moe" 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for i in 1 2; do $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; done
