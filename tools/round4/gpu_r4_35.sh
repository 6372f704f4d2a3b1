#!/bin/bash
# round 4, call 35: the whole GPU suite on the final tree (IQ4_XS added after call 31's evidence run)
O=gpurun_out/r4_35; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu_full.txt; grep -E "passed|failed" $O/pytest_gpu_full.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null | cut -c1-200 | tee $O/bench_line_steps20.json
