#!/bin/bash
# round 4, call 3: RMS_NORM debug, second A/B of the prologue order (PIPE off everywhere), partial offload / layer split tests, TP tests
O=gpurun_out/r4_3; mkdir -p $O
python tools/round4/rms_debug.py 2>&1 | grep -v "^\[rank" | tee $O/rms_debug.txt | tail -30
B="python bench.py --no-cpu-baseline --no-pmc"
run() { name=$1; lib=$PWD/chatllm.cpp_amd/libchatllm_hip$2.so
      CLLM_LIB=$lib $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps20  %.1f tok/s  tail %s' % ('$name', d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%-10s steps256 %.1f tok/s  tail %s  gate/up %.2f us' % ('$name', d['value'], d['greedy_tail'], d['roofline']['avg_us']))" | tee -a $O/summary.txt
      CLLM_LIB=$lib python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep -v "gate_up \|down  " | sed "s/^/$name /" | tee -a $O/summary.txt; }
run orig _orig
run base _base
run bar0 _bar0
run wait0 _wait0
run waitnb _waitnb
run orig2 _orig
CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip_bar0.so python tools/gemv_phase_probe.py 2>&1 | tee $O/phase_bar0.txt | tail -28
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "partial_offload" 2>&1 | tail -15 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_tp.py -m gpu -q -x 2>&1 | tail -5 | tee -a $O/summary.txt
