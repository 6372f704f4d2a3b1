#!/bin/bash
# round 5, call 11: a second step of weight prefetch issued IN THE MIDDLE of the prologue (GEMV_P=3 GEMV_PACE=1) against both steps at entry (p3) and one step (default)
O=gpurun_out/r5_11; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep -v "gate_up \|down  " | sed "s/^/$name /" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s steps20  %.1f tok/s  decode_512 %.1f  tail %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run p2 ""
run p3pace _p3pace
run p3 _p3
run p2 ""
run p3pace _p3pace
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_p3pace.so python tools/gemv_phase_probe.py 2>&1 | tee $O/phase_p3pace.txt | tail -30
