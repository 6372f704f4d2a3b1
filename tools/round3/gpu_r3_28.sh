#!/bin/bash
O=gpurun_out/r3ab; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mul_mat_id or other_formats or k_quants" 2>&1 | tail -4 | tee $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -k "mixtral or k_quants" 2>&1 | tail -4 | tee $O/pytest_dropin.txt
