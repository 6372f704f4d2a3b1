#!/bin/bash
# round 5, call 3: k_gemv_ldr with constant-only waits in the loader (steady state vmcnt(3 D), flush otherwise)
O=gpurun_out/r5_3; mkdir -p $O
for r in 0 2; do
  CLLM_GEMV_LDR=$r timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | sed "s/^/ldr=$r /" | tee -a $O/summary.txt
done
CLLM_GEMV_LDR=2 timeout 120 python tools/gemv_phase_probe.py --ring 2>&1 | tee $O/phase_ldr.txt | tail -30
CLLM_GEMV_LDR=2 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv" 2>&1 | tail -3 | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for r in 0 1; do
  CLLM_GEMV_LDR=$r timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ldr=$r steps20  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
