#!/bin/bash
# round 3, call 5: persistent decode launch with the fixed barrier + the hand-written dense fp16 GEMM
O=gpurun_out/r3e; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -12 | tee $O/pytest_llama.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "dense_f16" 2>&1 | tail -6 | tee $O/pytest_dense_f16.txt
timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline 2>$O/bench_persist.err | tee $O/bench_persist.json | cut -c1-300
CLLM_DECODE_PERSIST=0 timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline 2>$O/bench_5launch.err | tee $O/bench_5launch.json | cut -c1-300
timeout 300 python tools/persist_phase_probe.py 2>&1 | tail -9 | tee $O/persist_phases.txt
CLLM_PREFILL=f16 timeout 300 python tools/gemv_bench.py --types q4_0,q4_k,q8_0 --cols 4096 --iters 8 --shapes gate_up,down 2>&1 | grep -E "cols=" | sed "s/^/[f16] /" | tee $O/mmd.txt
for t in q4_0 q4_k; do CLLM_PREFILL=f16 timeout 300 python tools/prefill_bench.py --reps 3 --wtype $t 2>&1 | grep prefill | sed "s/^/[f16] /" | tee -a $O/prefill_bench.txt; done
