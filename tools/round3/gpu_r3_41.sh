#!/bin/bash
# round-end evidence: the whole GPU suite, then the bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r03_pytest_gpu_full.txt
tail -3 gpurun_out/r03_pytest_gpu_full.txt
timeout 900 python bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_err.txt
tail -c 1500 gpurun_out/r03_bench_line.json
