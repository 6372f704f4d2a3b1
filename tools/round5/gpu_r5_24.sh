#!/bin/bash
# round 5, call 24: k_attn_dec with glibc's exp2f table held in the lanes (requested at kernel entry) for the soft_max's n_kv mod 8 leftovers, against the build that
# fetches it from constant memory behind the maximum
O=gpurun_out/r5_24; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "attn or soft_max or rope" 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-prefill"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d.get('kernels',{}); print('%-6s steps20  %.1f tok/s  decode_512 %.1f  tail %s  attention %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail'], json.dumps(k.get('attention', k))[:200]))" | tee -a $O/summary.txt; }
run old _old
run new ""
run old _old
run new ""
