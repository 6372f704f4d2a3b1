#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r3w.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "team32 or rows32 or quant_gemv or vec_fused" 2>&1 | tail -3 >> $O
timeout 120 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 2>&1 | grep -E "gate_up|lm_head" >> $O
timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_0,q8_0 2>&1 | grep -E "gate_up|lm_head|qkv" >> $O
for t in q4_0 q8_0 q4_1; do
timeout 300 python bench.py --wtype $t --steps 512 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t decode', round(d['value'],1), 'tok/s')" >> $O
done
cat $O
