#!/bin/bash
# round 5, call 38: randomized differential run of the C ABI against the oracle on the final tree (22 weight types incl. the ternary and codebook formats, attention at random lengths)
O=gpurun_out/r5_38; mkdir -p $O
timeout 400 python tools/fuzz_parity.py --seconds 150 --seed 505 2>&1 | tail -12 | tee -a $O/summary.txt
