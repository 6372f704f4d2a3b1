#!/bin/bash
# round 4, call 13: several decode steps per hipGraph in the runner's greedy loop (CLLM_DECODE_GRAPH_STEPS = 1 | 4 | 8)
O=gpurun_out/r4_13; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels"
for rep in 1 2; do for n in 1 4 8; do
  CLLM_DECODE_GRAPH_STEPS=$n $B --steps 512 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph_steps=$n steps512 %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
  CLLM_DECODE_GRAPH_STEPS=$n $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('graph_steps=$n steps20  %.1f tok/s  tail %s' % (d['value'], d['greedy_tail']))" | tee -a $O/summary.txt
done; done
timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_golden.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/summary.txt
