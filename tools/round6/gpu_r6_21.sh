#!/bin/bash
# round 6, call 21: the gather prologue of the fused all-reduce at 1 / 2 / 4 / 8 ranks (granules all present): the review's "PRO 5 prologue stamp at 8 ranks"
O=gpurun_out/r6_21; mkdir -p $O
timeout 300 python tools/tp_gather_probe.py 2>&1 | tee $O/tp_gather_probe.txt
