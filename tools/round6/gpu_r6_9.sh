#!/bin/bash
# round 6, call 9: (a) the tensor-parallel device: fallback cases (sparse MoE, partial offload, -fa) + ranks on streams of their own; (b) what the decode-ahead's separate stream
# submissions cost per token (tools/micro/submission_gap.hip)
O=gpurun_out/r6_9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel" -s 2>&1 | tail -25 | tee $O/pytest_tp.txt
for i in 1 2; do timeout 120 tools/micro/bin/submission_gap 2>&1 | tee -a $O/submission_gap.txt; done
