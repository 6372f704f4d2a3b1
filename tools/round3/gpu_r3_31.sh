#!/bin/bash
O=gpurun_out/r3ae; mkdir -p $O
timeout 1500 bash tools/dropin_mixtral.sh 2>&1 | tail -12 | tee $O/cfg5_mixtral.txt
