#!/bin/bash
mkdir -p gpurun_out/r2e
timeout 2000 python -m pytest tests -m gpu -q --timeout=1500 -x > gpurun_out/r2e/pytest_all.log 2>&1; tail -8 gpurun_out/r2e/pytest_all.log | cut -c1-300
timeout 900 python bench.py > gpurun_out/r2e/bench.json 2> gpurun_out/r2e/bench.err; tail -c 3000 gpurun_out/r2e/bench.json; tail -5 gpurun_out/r2e/bench.err
