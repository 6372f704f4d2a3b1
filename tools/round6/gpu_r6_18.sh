#!/bin/bash
# round 6, call 18: tensor parallel behind the boundary at BASELINE cfg4's and cfg2's block shapes (8 ranks, 2 layers each)
O=gpurun_out/r6_18; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "real_block_shapes" -s 2>&1 | tail -25 | tee $O/pytest_tp_big.txt
