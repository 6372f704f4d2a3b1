#!/bin/bash
# round 6, call 1: (a) the two new driver-visible full-depth tests + the drop-in suite's fast cases on the fresh build; (b) the fast prefill mode decomposed at BASELINE cfg3
# (4096-token prompt, 8B shapes): int8-MFMA mat-muls + EXACT attention, and EXACT mat-muls + flash attention, against the CPU host -- Q4_0 and Q4_K
O=gpurun_out/r6_1; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "full_depth_80 or full_depth_32 or cpu_vs_our_module" 2>&1 | tail -5 | tee $O/pytest_depth.txt
for wt in q4_0 q4_k; do
  timeout 1500 python tools/long_prompt_parity.py --wtype $wt --n-prompt 4096 --n-dec 8 --threads 64 --modes "default,fast,mmq+exact-attn,exact-mm+flash,f16,f16+exact-attn" 2>&1 | tee -a $O/prefill_mode_decomposition.txt
done
