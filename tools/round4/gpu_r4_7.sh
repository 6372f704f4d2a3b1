#!/bin/bash
# round 4, call 7: two workgroups per CU for the decode mat-vec (8 waves per SIMD)?
O=gpurun_out/r4_7; mkdir -p $O
for w in 1 2; do
  CLLM_GEMV_WGS_PER_CU=$w python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused | grep -v "gate_up \|down  " | sed "s/^/wgs_per_cu=$w /" | tee -a $O/summary.txt
  CLLM_GEMV_WGS_PER_CU=$w python bench.py --no-cpu-baseline --no-pmc --no-kernels --steps 128 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgs_per_cu=$w steps128 %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done
