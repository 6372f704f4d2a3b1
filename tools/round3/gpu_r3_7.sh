#!/bin/bash
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_tp.py -q -x -k two_ranks 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" | tail -40 | cut -c1-400 | tee $O/pytest_tp.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -q -x -k "persistent" 2>&1 | tail -5 | tee $O/pytest_persist.txt
for mode in f16 fast exact; do CLLM_PREFILL=$mode timeout 300 python tools/gemv_bench.py --types q4_0,q4_k --cols 4096 --iters 6 --shapes gate_up,down 2>&1 | grep -E "cols=" | sed "s/^/[$mode] /" | tee -a $O/gemm_tile_order.txt; done
for mode in f16 fast exact; do CLLM_PREFILL=$mode timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep prefill | sed "s/^/[$mode] /" | tee -a $O/prefill_bench.txt; done
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mul_mat" 2>&1 | tail -3 | tee $O/pytest_ops.txt
