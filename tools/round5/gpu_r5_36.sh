#!/bin/bash
# round 5, call 36+: TQ1_0 / TQ2_0 (ternary) and IQ2_XXS / IQ2_XS / IQ2_S / IQ3_XXS / IQ3_S + IQ1_S / IQ1_M (codebook) weights: mat-mul at every column count, MUL_MAT_ID, GET_ROWS against the oracle; model files in both formats through the unmodified host
O=gpurun_out/r5_36; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "other_formats or mul_mat_id or get_rows" 2>&1 | grep -E "passed|failed|error|assert|Error" | tail -4 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "other_formats" 2>&1 | grep -E "passed|failed|error|assert|Error" | tail -4 | tee -a $O/summary.txt
