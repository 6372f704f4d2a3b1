#!/bin/bash
# round 5, call 35: the order proof of the soft_max's double total in EVERY soft_max kernel (serial fallback); a library built with SOFT_FORCE_SERIAL=1 (always the serial
# order) runs the same tests; the MoE launch and the decode line must not move
O=gpurun_out/r5_35; mkdir -p $O
for v in "" _serial; do
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or soft_max or rope or moe or mul_mat_id or composite or flash" 2>&1 | grep -E "passed|failed|error|assert" | tail -3 | sed "s/^/lib$v ops: /" | tee -a $O/summary.txt
  CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$v.so timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2 | sed "s/^/lib$v llama: /" | tee -a $O/summary.txt
done
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "mixtral or long" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
timeout 300 python tools/moe_bench.py --iters 400 2>&1 | grep "router_gate_up\|fold+down" | tee -a $O/summary.txt
python bench.py --no-cpu-baseline --no-pmc --no-prefill --no-kernels --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
