#!/bin/bash
# LDS-DMA weight stream (gemv_rows.hip) for every Q4_K decode mat-vec, with and without the non-temporal policy on the DMA loads (MI355X_MICROARCH.md nt-weights)
O=gpurun_out/r3af; mkdir -p $O
TAG=$1
for mode in 1 2 8 4; do
  CLLM_GEMV_ROWS=$mode timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 64 2>&1 | grep fused | sed "s/^/[$TAG CLLM_GEMV_ROWS=$mode] /" | tee -a $O/gemv_rows_$TAG.txt
done
for mode in 1 2; do
  CLLM_GEMV_ROWS=$mode timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$TAG CLLM_GEMV_ROWS=$mode] decode', round(d['value'],1), 'tok/s')" | tee -a $O/gemv_rows_$TAG.txt
done
