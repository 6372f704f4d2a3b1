#!/bin/bash
# round 3, call 1: MFMA layout probe + where the long-prompt parity stands at real shapes (round 2's kernels)
O=gpurun_out/r3a; mkdir -p $O
tools/micro/bin/mfma_probe 2>&1 | tee $O/mfma_probe.txt
timeout 900 python tools/long_prompt_parity.py --wtype q4_0 --n-prompt 512 --modes default,exact-r02 2>&1 | tee $O/long_prompt_512_q4_0.txt
timeout 900 python tools/long_prompt_parity.py --wtype q4_k --n-prompt 512 --modes default,exact-r02 2>&1 | tee $O/long_prompt_512_q4_k.txt
timeout 1500 python tools/long_prompt_parity.py --wtype q4_0 --n-prompt 4096 --n-dec 4 --modes default,exact-r02 2>&1 | tee $O/long_prompt_4096_q4_0.txt
