#!/bin/bash
# team kernel: parity, then per-launch times with it off / on, then the decode bench of the three 32-block types
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "team32" 2>&1 | tail -15 > gpurun_out/team32_tests.txt
tail -3 gpurun_out/team32_tests.txt
for m in 0 1; do
  CLLM_GEMV_TEAM32=$m timeout 300 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 > gpurun_out/team32_gemv_$m.txt 2>&1
  CLLM_GEMV_TEAM32=$m timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_0,q8_0 >> gpurun_out/team32_gemv_$m.txt 2>&1
done
for t in q4_0 q8_0; do for m in 0 1; do
  CLLM_GEMV_TEAM32=$m timeout 300 python bench.py --wtype $t --steps 128 --warmup 8 --no-cpu-baseline --no-pmc 2>&1 | tail -1 > gpurun_out/team32_bench_${t}_$m.txt
done; done
head -c 600 gpurun_out/team32_bench_q4_0_0.txt; echo; head -c 600 gpurun_out/team32_bench_q4_0_1.txt
