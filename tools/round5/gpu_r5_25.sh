#!/bin/bash
# round 5, call 25: k_attn_dec phase stamps at cached lengths with / without soft_max leftovers (n mod 8) and V.P leftovers (n mod 32), old and new build
O=gpurun_out/r5_25; mkdir -p $O
for n in 288 296 300 319; do
  for v in _old ""; do
    echo "== lib$v n_ctx $n" | tee -a $O/summary.txt
    CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 200 python tools/attn_phase_probe.py $n 2>&1 | tail -5 | tee -a $O/summary.txt
  done
done
