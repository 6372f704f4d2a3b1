#!/bin/bash
# round 3, call 3: where mmx.hip's time goes -- instruction rates, occupancy, A/B builds (scalar fma, staging only, compute only)
O=gpurun_out/r3c; mkdir -p $O
tools/micro/bin/mfma_rate 2>&1 | tee $O/mfma_rate.txt
for lib in "" _mmx_scalar _mmx_nocompute _mmx_nostage; do
  CLLM_DEBUG=1 CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/gemv_bench.py --types q4_0 --cols 4096 --iters 4 --shapes gate_up,down 2>&1 | grep -E "q4_0|occupancy" | sed "s/^/[mmx$lib] /" | tee -a $O/mmx_variants.txt
done
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip.so timeout 300 python tools/gemv_bench.py --types q4_k,q8_0 --cols 4096 --iters 4 --shapes gate_up 2>&1 | grep -E "q4_k|q8_0" | tee -a $O/mmx_variants.txt
CLLM_PREFILL=fast timeout 300 python tools/gemv_bench.py --types q4_0,q4_k --cols 4096 --iters 4 --shapes gate_up 2>&1 | grep -E "q4_" | sed "s/^/[fast mmq] /" | tee -a $O/mmx_variants.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -q -x -k "long_prompts" 2>&1 | tail -5 | tee $O/pytest_llama_long.txt
