#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/team32_dbg.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "team32 or rows32 or quant_gemv" 2>&1 | tail -3 >> gpurun_out/team32_dbg.txt
for tm in 1; do echo "== team $tm all" >> gpurun_out/team32_dbg.txt
  CLLM_GEMV_TEAM32=$tm timeout 120 python tools/gemv_bench.py --fused --types q4_1 2>&1 | grep -E " o |down|qkv" >> gpurun_out/team32_dbg.txt
done
for t in q4_0 q8_0; do for m in 0 1; do
  CLLM_GEMV_TEAM32=$m timeout 300 python bench.py --wtype $t --steps 128 --warmup 8 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | cut -c1-120 >> gpurun_out/team32_dbg.txt
done; done
timeout 900 python -m pytest tests/test_gpu_llama.py -x -q -m gpu 2>&1 | tail -3 >> gpurun_out/team32_dbg.txt
cat gpurun_out/team32_dbg.txt
