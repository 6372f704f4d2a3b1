#!/bin/bash
# round 5, call 22: EPI 5 (router inside the experts' gate / up launch) with the wave's router row requested at kernel entry, against the build without it
O=gpurun_out/r5_22; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "moe or mul_mat_id" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
for r in 1 2; do
  for v in nopre base; do
    lib=$PWD/chatllm.cpp_amd/libchatllm_hip.so; [ $v = nopre ] && lib=$PWD/chatllm.cpp_amd/libchatllm_hip_nopre.so
    CLLM_LIB=$lib timeout 300 python tools/moe_bench.py --iters 400 2>&1 | grep "router_gate_up\|fold+down" | sed "s/^/$v /" | tee -a $O/summary.txt
  done
done
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "mixtral" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
