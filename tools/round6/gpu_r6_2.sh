#!/bin/bash
# round 6, call 2: the fused FFN launch (ffn_fused.hip) -- parity tests, the block's time against the two launches and against the null hand-off, the decode step with it on / off
O=gpurun_out/r6_2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_llama.py -x -q -m gpu -k "fused_ffn or fused_decode_path or greedy" 2>&1 | tail -15 | tee $O/pytest_ffn.txt
timeout 300 python tools/ffn_bench.py 2>&1 | tee $O/ffn_bench.txt
for m in 0 1 2; do
  for s in 20 256; do
    CLLM_FFN_FUSED=$m timeout 400 python bench.py --steps $s --warmup 5 --no-pmc --no-kernels --no-prefill --no-cpu-baseline 2>$O/bench_err_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CLLM_FFN_FUSED=$m steps $s: %.1f tok/s  %.4f ms/step  tail %s' % (d['value'], d['ms_per_step'], d.get('greedy_tail')))" | tee -a $O/decode_ab.txt
  done
done
