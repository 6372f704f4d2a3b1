#!/bin/bash
# A/B the decode mat-vec builds: tools/ab_gemv.sh name1 name2 ...   ("base" = libchatllm_hip.so)
mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=chatllm.cpp_amd/libchatllm_hip_$v.so; [ "$v" = base ] && lib=chatllm.cpp_amd/libchatllm_hip.so
  echo "== $v" | tee -a gpurun_out/ab/ab.log
  CLLM_LIB=$PWD/$lib python tools/gemv_bench.py --fused --types ${TYPES:-q4_k} --iters 128 2>&1 | grep fused | tee -a gpurun_out/ab/ab.log
done
