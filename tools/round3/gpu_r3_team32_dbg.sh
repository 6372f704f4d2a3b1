#!/bin/bash
mkdir -p gpurun_out; : > gpurun_out/team32_dbg.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "team32 or rows32 or quant_gemv" 2>&1 | tail -3 >> gpurun_out/team32_dbg.txt
timeout 300 python tools/team32_phase_probe.py 2>&1 | grep -E "avg|emit|chain|barrier" >> gpurun_out/team32_dbg.txt 2>&1
timeout 120 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 2>&1 | grep -E " o |down|qkv" >> gpurun_out/team32_dbg.txt
timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_0,q8_0 2>&1 | grep -E " o |down|qkv" >> gpurun_out/team32_dbg.txt
timeout 300 python bench.py --wtype q4_0 --steps 128 --warmup 8 --no-cpu-baseline --no-pmc 2>&1 | tail -1 | cut -c1-120 >> gpurun_out/team32_dbg.txt
cat gpurun_out/team32_dbg.txt
