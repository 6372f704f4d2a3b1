#!/bin/bash
# round 4, call 14: k_mmx with two LDS stages and one barrier per K step (Q4_0 / Q8_0): exactness + times
O=gpurun_out/r4_14; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "mul_mat" 2>&1 | tail -3 | tee $O/pytest.txt
for lib in _r03 ""; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/gemv_bench.py --types q4_0,q8_0 --cols 4096 --iters 4 --shapes qkv,gate_up,down 2>&1 | grep -E "q4_0|q8_0" | sed "s/^/[lib${lib:-_r04}] /" | tee -a $O/mmx_dbuf.txt
done
timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | tee -a $O/mmx_dbuf.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x -k "long_prompt" 2>&1 | tail -2 | tee -a $O/pytest.txt
