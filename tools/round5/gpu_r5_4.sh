#!/bin/bash
# round 5, call 4: the 16-values-per-lane prologue quantizer (quant16_q8_K) in k_gemv_dec: q16_0 = four values per lane (round 4), q16_1 = plain-quantize prologue only, default = + RMS_NORM prologue
O=gpurun_out/r5_4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv or rms or mul_mat_id or moe or packed" 2>&1 | tail -5 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep -v "gate_up \|down  " | sed "s/^/$name /" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-8s steps256 %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt; }
run q16_0 _q16_0
run q16_1 _q16_1
run q16_2 ""
run q16_0 _q16_0
run q16_2 ""
python tools/gemv_phase_probe.py 2>&1 | tee $O/phase_q16.txt | tail -30
