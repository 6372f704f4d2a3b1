#!/bin/bash
mkdir -p gpurun_out/r2h
python tools/prefill_modes_probe.py 2>&1 | tee gpurun_out/r2h/prefill_modes.txt
python tools/prefill_bench.py --reps 3 2>&1 | grep prefill | tee gpurun_out/r2h/prefill_bench.txt
CLLM_PREFILL=f16 python tools/prefill_bench.py --reps 3 2>&1 | grep prefill | tee -a gpurun_out/r2h/prefill_bench.txt
CLLM_PREFILL=f16 python tools/prefill_bench.py --reps 3 --wtype q4_k 2>&1 | grep prefill | tee -a gpurun_out/r2h/prefill_bench.txt
