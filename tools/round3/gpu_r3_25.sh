#!/bin/bash
O=gpurun_out/r3y; mkdir -p $O
CLLM_PREFILL=f16 timeout 300 python -m pytest tests/test_gpu_ops.py -q -x -k "dense_f16" 2>&1 | tail -3 | tee $O/pytest_f16.txt
CLLM_PREFILL=f16 timeout 300 python tools/gemv_bench.py --types q4_0,q4_k,q8_0 --cols 4096 --iters 8 --shapes qkv,o,gate_up,down 2>&1 | grep -E "K=" | sed "s/^/[f16, X fragments from global] /" | tee $O/f16_gemm.txt
CLLM_PREFILL=f16 timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | tee -a $O/f16_gemm.txt
