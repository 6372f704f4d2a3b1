#!/bin/bash
# round 4, call 10: did the fast / f16 prefill modes get slower?  round 3's library vs this round's, same box, same process order
O=gpurun_out/r4_10; mkdir -p $O
for lib in _r03 "" _r03 ""; do
  for mode in fast f16 exact; do
    CLLM_PREFILL=$mode CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[lib${lib:-_r04} $mode] /" | cut -c1-120 | tee -a $O/prefill_ab.txt
  done
done
