#!/bin/bash
# round 6, call 23: randomized differential run of the C ABI against the oracle on the final tree (a new seed)
O=gpurun_out/r6_23; mkdir -p $O
timeout 500 python tools/fuzz_parity.py --seconds 240 --seed 606 2>&1 | tail -12 | tee $O/summary.txt
