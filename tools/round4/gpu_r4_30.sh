#!/bin/bash
# round 4, call 30: the serial fallback as a CALLED function again (out of the decode kernels' cold straight-line code: inlined it cost 1 %), now on wave 0's lanes
# (few registers: no call frame, no spills) -- against the previous build
O=gpurun_out/r4_30; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "rms or norm_prologues" 2>&1 | tail -2 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 512 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps512 %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run head _head; run new ""; run head _head; run new ""
