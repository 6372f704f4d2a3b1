#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r3z.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dense_f16" 2>&1 | tail -2 >> $O
CLLM_PREFILL=f16 timeout 300 python tools/gemv_bench.py --types q4_0,q4_k,q8_0 --cols 4096 --iters 8 --shapes qkv,o,gate_up,down 2>&1 | grep -E "K=" | sed "s/^/[f16, tile picked] /" >> $O
timeout 600 python bench.py --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); m=d['prefill']['modes']
print('cfg3 prefill:', {k: round(v['ms'],1) for k,v in m.items()})" >> $O
cat $O
