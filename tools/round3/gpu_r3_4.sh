#!/bin/bash
# round 3, call 4: persistent all-layers decode launch (parity, speed, phase stamps) + the reworked mmx.hip
O=gpurun_out/r3d; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -12 | tee $O/pytest_llama.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "mul_mat" 2>&1 | tail -4 | tee $O/pytest_ops_mul_mat.txt
timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline 2>$O/bench_persist.err | tee $O/bench_persist.json
CLLM_DECODE_PERSIST=0 timeout 300 python bench.py --steps 128 --warmup 16 --no-cpu-baseline 2>$O/bench_5launch.err | tee $O/bench_5launch.json
timeout 300 python tools/persist_phase_probe.py 2>&1 | tail -9 | tee $O/persist_phases.txt
timeout 300 python tools/gemv_bench.py --types q4_0,q4_k,q8_0 --cols 4096 --iters 4 --shapes gate_up,down 2>&1 | grep -E "cols=" | tee $O/mmx_v2.txt
for mode in exact fast; do CLLM_PREFILL=$mode timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep prefill | sed "s/^/[$mode] /" | tee -a $O/prefill_bench.txt; done
