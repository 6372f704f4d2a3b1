#!/bin/bash
# round 4, call 29: the order-exact RMS_NORM's serial fallback on wave 0's lanes (no called function: no scratch in the decode mat-vec kernels) against the
# previous build (libchatllm_hip_head.so = the committed tree before it) -- parity tests, decode tok/s, the prompt in its three modes
O=gpurun_out/r4_29; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "rms or norm_prologues or fused or quant_gemv or fuzz" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_fattn.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      for rep in 1 2; do
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 512 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps512 %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      done; }
run head _head
run new ""
for lib in _head ""; do
  for mode in exact fast f16; do
    echo "lib=${lib:-new} CLLM_PREFILL=$mode" | tee -a $O/summary.txt
    CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so CLLM_PREFILL=$mode timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | tee -a $O/summary.txt
  done
done
