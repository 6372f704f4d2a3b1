#!/bin/bash
# round 6, call 4: fused FFN, final form of the experiment (ring filled behind the prologue barrier; the edge polled in its FIFO slot)
O=gpurun_out/r6_4; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_llama.py -x -q -m gpu -k "fused_ffn" 2>&1 | tail -5 | tee $O/pytest_ffn.txt
timeout 300 python tools/ffn_bench.py 2>&1 | tee $O/ffn_bench.txt
for m in 0 1; do
    CLLM_FFN_FUSED=$m timeout 400 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --no-cpu-baseline 2>$O/bench_err_$m.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CLLM_FFN_FUSED=$m steps 20: %.1f tok/s  %.4f ms/step  tail %s' % (d['value'], d['ms_per_step'], d.get('greedy_tail')))" | tee -a $O/decode_ab.txt
done
