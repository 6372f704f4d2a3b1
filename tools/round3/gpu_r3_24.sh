#!/bin/bash
O=gpurun_out/r3x; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -16 | tee $O/pytest_gpu_full.txt
