#!/bin/bash
# round 4, call 36: k_mmx staging the activations from an fp16 copy made once per mat-mul (CLLM_MMX_X16=1, Q4_0 / Q8_0, prompts) against converting them in every row tile
O=gpurun_out/r4_36; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "quant_exact_many_columns or quant_gemm" 2>&1 | tail -2 | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -2 | tee -a $O/summary.txt
for x in 0 1; do
  echo "CLLM_MMX_X16=$x" | tee -a $O/summary.txt
  CLLM_MMX_X16=$x timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | cut -c1-120 | tee -a $O/summary.txt
done
