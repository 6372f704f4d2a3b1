#!/bin/bash
# round 4, call 4: RMS_NORM debug, partial offload / layer split tests, TP tests (uneven split), rms tests
O=gpurun_out/r4_4; mkdir -p $O
python tools/round4/rms_debug.py 2>&1 | grep -v "^\[rank" | tee $O/rms_debug.txt | tail -30
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "rms or norm_prologues" 2>&1 | tail -8 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "partial_offload" 2>&1 | tail -25 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -q 2>&1 | tail -15 | tee -a $O/summary.txt
python bench.py --no-cpu-baseline --no-pmc --steps 20 --warmup 5 2>/dev/null | cut -c1-400 | tee -a $O/summary.txt
