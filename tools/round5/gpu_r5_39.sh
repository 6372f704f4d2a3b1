#!/bin/bash
# round 5, call 39: k_mmx (exact prefill GEMM, 32-block types) with a 16 x 32 patch per wave (MMX_MI=1: 64 accumulator registers, 156 VGPRs) at three waves per SIMD against the 32 x 32 patch at two
O=gpurun_out/r5_39; mkdir -p $O
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_mi1o3.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "mul_mat" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
for lib in "" _mi1o3 "" _mi1o3; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/gemv_bench.py --types q4_0,q8_0 --cols 4096 --iters 4 --shapes gate_up,down 2>&1 | grep -E "q4_0|q8_0" | sed "s/^/[lib${lib:-_base}] /" | tee -a $O/summary.txt
done
