#!/bin/bash
# round 4, call 5: k_mmx A/B (next block's LDS reads under the last patch's folds) + time decomposition, exactness tests of the changed kernels, bench with the new sections
O=gpurun_out/r4_5; mkdir -p $O
for lib in "" _mmx_old _mmx_nostage _mmx_nocompute _mmx_scalar; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/gemv_bench.py --types q4_0 --cols 4096 --iters 4 --shapes gate_up,down 2>&1 | grep -E "q4_0" | sed "s/^/[mmx$lib] /" | tee -a $O/mmx_variants.txt
done
timeout 300 python tools/gemv_bench.py --types q8_0,q4_1,q4_k --cols 4096 --iters 4 --shapes gate_up 2>&1 | grep -E "q8_0|q4_1|q4_k" | sed "s/^/[mmx] /" | tee -a $O/mmx_variants.txt
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_mmx_old.so timeout 300 python tools/gemv_bench.py --types q8_0,q4_1 --cols 4096 --iters 4 --shapes gate_up 2>&1 | grep -E "q8_0|q4_1" | sed "s/^/[mmx_old] /" | tee -a $O/mmx_variants.txt
timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | grep -v "^\[rank" | tail -3 | tee $O/prefill.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "mul_mat or rope or rms" 2>&1 | tail -4 | tee $O/pytest_ops.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x -k "long_prompt" 2>&1 | tail -3 | tee -a $O/pytest_ops.txt
python bench.py --no-cpu-baseline --no-pmc --no-prefill --steps 128 --warmup 16 2>/dev/null | tee $O/bench_line.json | cut -c1-1800
