#!/bin/bash
# round 5, call 16: RMS_NORM prologue with ONE wave deriving the scale (GEMV_SCALE1=1) vs all sixteen
O=gpurun_out/r5_16; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep "qkv\|gate_up_silu\|lm_head" | sed "s/^/$name /" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s steps20  %.1f tok/s  decode_512 %.1f  tail %s' % ('$name', d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run base ""
run sc1 _sc1
run base ""
run sc1 _sc1
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_sc1.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "norm_prologues or fused or rms" 2>&1 | tail -3 | tee -a $O/summary.txt
