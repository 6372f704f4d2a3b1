#!/bin/bash
# round 5, call 31: k_attn_dec's soft_max done by EVERY wave for itself up to 512 cached positions (no workgroup barrier between the scores and V.P) against the build before;
# the always-serial-order build runs the same tests
O=gpurun_out/r5_31; mkdir -p $O
for v in "" _serial; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or soft_max or rope" 2>&1 | grep -E "passed|failed|error|assert" | tail -3 | sed "s/^/lib$v: /" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
for n in 48 288 300 500; do
  for v in _old ""; do
    echo "== lib$v n_ctx $n" | tee -a $O/summary.txt
    CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 200 python tools/attn_phase_probe.py $n 2>&1 | tail -3 | tee -a $O/summary.txt
  done
done
B="python bench.py --no-cpu-baseline --no-pmc --no-prefill --no-kernels"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-6s steps20  %.1f tok/s  decode_512 %.1f  tail %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run old _old
run new ""
run old _old
run new ""
