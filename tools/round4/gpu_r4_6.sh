#!/bin/bash
# round 4, call 6: VALU issue rates by occupancy and beside MFMAs (tools/micro/valu_rate.hip, built by: hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/valu_rate tools/micro/valu_rate.hip)
mkdir -p gpurun_out/r4_6; tools/micro/bin/valu_rate | tee gpurun_out/r4_6/valu_rate.txt
