#!/bin/bash
O=gpurun_out/r3ag; mkdir -p $O
timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 3 2>&1 | tail -25 | tee $O/fuzz.txt
