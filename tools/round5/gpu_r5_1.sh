#!/bin/bash
# round 5, call 1: k_gemv_ring (the weight stream through a per-wave LDS-DMA ring, three steps in flight from kernel entry) against k_gemv_dec:
# per launch (HIP events), in-kernel stamps, the decode step at 20 / 256 steps, and the bit-identity tests on the ring path
O=gpurun_out/r5_1; mkdir -p $O
for r in 0 3 2; do
  CLLM_GEMV_RING=$r python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | sed "s/^/ring=$r /" | tee -a $O/summary.txt
done
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
for r in 0 3 2 0 3; do
  CLLM_GEMV_RING=$r $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring=$r steps20  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
  CLLM_GEMV_RING=$r $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring=$r steps256 %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
CLLM_GEMV_RING=3 python tools/gemv_phase_probe.py --ring 2>&1 | tee $O/phase_ring3.txt | tail -30
CLLM_GEMV_RING=0 python tools/gemv_phase_probe.py 2>&1 | tee $O/phase_dec.txt | tail -30
CLLM_GEMV_RING=3 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "fused or norm_prologues or quant_gemv or no_scratch" 2>&1 | tail -5 | tee -a $O/summary.txt
CLLM_GEMV_RING=3 timeout 600 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
