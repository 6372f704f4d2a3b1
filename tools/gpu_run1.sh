#!/bin/bash
# GPU batch 1: exactness probe, full GPU test suite, short bench per weight type
mkdir -p gpurun_out/r2a
python tools/exact_probe.py > gpurun_out/r2a/exact_probe.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 --deselect tests/test_gpu_dropin.py > gpurun_out/r2a/pytest_all.log 2>&1
for t in q4_k q4_0 q8_0 q4_1; do timeout 300 python bench.py --steps 128 --wtype $t --no-cpu-baseline > gpurun_out/r2a/bench_$t.log 2>&1; done
tail -3 gpurun_out/r2a/exact_probe.log; tail -3 gpurun_out/r2a/pytest_all.log; cat gpurun_out/r2a/bench_q4_k.log | tail -1 | cut -c1-400
