#!/bin/bash
# round 5, call 28: k_attn_dec as smaller code (13.7 -> 8.0 KB: one branch-free loop body each for the scores and for V.P, the new row's score by one lane group): tests, stamps, decode A/B
O=gpurun_out/r5_28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or soft_max or rope" 2>&1 | grep -E "passed|failed|error|assert" | tail -4 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
for n in 48 128 288 300 319 500; do
  for v in _old ""; do
    echo "== lib$v n_ctx $n" | tee -a $O/summary.txt
    CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 200 python tools/attn_phase_probe.py $n 2>&1 | tail -4 | tee -a $O/summary.txt
  done
done
B="python bench.py --no-cpu-baseline --no-pmc --no-prefill --no-kernels"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s steps20  %.1f tok/s  decode_512 %.1f  tail %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run old _old
run new ""
run old _old
run new ""
