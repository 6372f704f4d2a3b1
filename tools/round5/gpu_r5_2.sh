#!/bin/bash
# round 5, call 2: k_gemv_ldr (wave 15 of the workgroup is a loader that fills per-consumer LDS rings by LDS-DMA from kernel entry; 15 consumer waves) against k_gemv_dec
O=gpurun_out/r5_2; mkdir -p $O
CLLM_GEMV_LDR=2 timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv" 2>&1 | tail -5 | tee -a $O/summary.txt
for r in 0 2; do
  CLLM_GEMV_LDR=$r timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | sed "s/^/ldr=$r /" | tee -a $O/summary.txt
done
CLLM_GEMV_LDR=2 timeout 120 python tools/gemv_phase_probe.py --ring 2>&1 | tee $O/phase_ldr.txt | tail -30
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for r in 0 1 2 0 1; do
  CLLM_GEMV_LDR=$r timeout 300 $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ldr=$r steps20  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
  CLLM_GEMV_LDR=$r timeout 300 $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ldr=$r steps256 %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
CLLM_GEMV_LDR=2 timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
