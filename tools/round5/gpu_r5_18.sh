#!/bin/bash
# round 5, call 18: (a) the runner / tensor-parallel tests on the folded greedy step, output kept; (b) the sparse-MoE launches at Mixtral shapes, per launch
O=gpurun_out/r5_18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_tp.py -m gpu -q -x > $O/pytest.txt 2>&1; echo "pytest rc $?" | tee -a $O/summary.txt
grep -E "passed|failed|error" $O/pytest.txt | tail -3 | tee -a $O/summary.txt
timeout 600 python tools/moe_bench.py --iters 400 2>&1 | tee -a $O/summary.txt
timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | tee -a $O/summary.txt
