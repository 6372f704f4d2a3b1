#!/bin/bash
# round 4, call 2: A/B of the decode mat-vec builds (orig = HEAD's kernel; base = kernel-argument trims; default = + pipelined down quantization;
# bar = + barrier between the activation and the weight requests; barp3 / barp5 / p3 = deeper weight prefetch) + the order-exact RMS_NORM tests
O=gpurun_out/r4_2; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps256 %.1f tok/s  tail %s  gate/up %.2f us' % ('$name', d['value'], d['greedy_tail'], d['roofline']['avg_us']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | sed "s/^/$name /" | tee -a $O/summary.txt; }
run orig _orig
run base _base
run pipe ""
run bar _bar
run barp3 _barp3
run barp5 _barp5
run p3 _p3
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "rms or norm_prologues or fused or quant_gemv" 2>&1 | tail -5 | tee -a $O/summary.txt
timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
python tools/gemv_phase_probe.py 2>&1 | tee $O/phase_pipe.txt | tail -30
