#!/bin/bash
# round 5, call 30: k_attn_dec's n mod 8 leftovers through glibc's expf INLINE with the exp2f table in LDS (fetched by half a wave with the first loads) -- against the call + constant-memory load
O=gpurun_out/r5_30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or soft_max or rope" 2>&1 | grep -E "passed|failed|error|assert" | tail -3 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
for n in 48 288 300 319; do
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
