#!/bin/bash
# crossover between the exact mat-vec in column chunks (mmvq, <= 4 columns per launch) and the exact GEMM (mmx) for short prompts
O=gpurun_out/r3u; mkdir -p $O
for cols in 2 4 5 8 12 16 24 32; do
  for thr in 33 2; do
    CLLM_MMQ_MIN_COLS=$thr timeout 120 python tools/gemv_bench.py --types q4_0,q4_k --cols $cols --iters 8 --shapes qkv,gate_up,down 2>&1 | grep -E "gate_up|down|qkv" | sed "s/^/[min_cols $thr] /" | tee -a $O/short_prompt_crossover.txt
  done
done
