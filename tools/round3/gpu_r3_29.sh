#!/bin/bash
O=gpurun_out/r3ac; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 1200 python -m pytest tests/test_gpu_dropin.py -q -x -k "other_formats or q6_k_and_q5_k_experts or other_k_quants" 2>&1 | tail -8 | tee $O/pytest_dropin.txt
