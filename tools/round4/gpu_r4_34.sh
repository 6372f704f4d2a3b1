#!/bin/bash
# round 4, call 34: IQ4_XS (gemv_kq.hip, dequant.h): mat-mul for every column count, MUL_MAT_ID, GET_ROWS against the oracle; a pure IQ4_XS model file through the
# unmodified reference host (CPU run vs -ngl all); the 30-s fuzz run with the type in its pool
O=gpurun_out/r4_34; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "other_formats or mul_mat_id or get_rows or fuzz or dequant" 2>&1 | tail -3 | tee $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "other_formats" 2>&1 | tail -3 | tee -a $O/summary.txt
