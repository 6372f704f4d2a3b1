#!/bin/bash
# round 4, call 16: k_mmx with a 64 x 32 tile (MI = 1: 64 accumulator registers, three workgroups per CU) against the 64 x 64 tile
O=gpurun_out/r4_16; mkdir -p $O
for lib in "" _mmx_n12 "" _mmx_n12; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/gemv_bench.py --types q4_0 --cols 4096 --iters 4 --shapes qkv,o,gate_up,down 2>&1 | grep -E "q4_0" | sed "s/^/[lib${lib:-_64x64}] /" | tee -a $O/mmx_tile.txt
done
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_mmx_n12.so timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[n12] /" | tee -a $O/mmx_tile.txt
timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[64x64] /" | tee -a $O/mmx_tile.txt
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_mmx_n12.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "mul_mat" 2>&1 | tail -2 | tee $O/pytest.txt
