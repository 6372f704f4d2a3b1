#!/bin/bash
O=gpurun_out/r3ai; mkdir -p $O
for m in 1 0 2 4; do CLLM_GEMV_ROWS32=$m timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q8_0 --iters 32 --shapes down_q,o 2>&1 | grep fused | sed "s/^/[q72 rows32 mode $m] /" | tee -a $O/q72_down.txt; done
timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_k --iters 32 2>&1 | grep fused | sed "s/^/[q72] /" | tee -a $O/q72_down.txt
