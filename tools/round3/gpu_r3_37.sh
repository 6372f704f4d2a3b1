#!/bin/bash
# after the team kernel: decode of the 32-block types (Llama-3-8B shapes), the per-launch times with the team kernel off / on, cfg4 on one GPU (runner)
O=gpurun_out/r3t; mkdir -p $O
for t in q4_0 q4_1 q8_0; do
  timeout 300 python bench.py --wtype $t --steps 512 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t decode', round(d['value'],1), 'tok/s, roofline frac', round(d['roofline']['frac'],3), d['roofline'].get('kernel',''))" | tee -a $O/decode.txt
done
for m in 0 1; do
  CLLM_GEMV_TEAM32=$m timeout 300 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 --iters 64 2>&1 | grep fused | sed "s/^/[team32 $m] /" | tee -a $O/gemv.txt
  CLLM_GEMV_TEAM32=$m timeout 300 python tools/gemv_bench.py --fused --model qwen2-72b --types q4_0,q8_0 --iters 64 2>&1 | grep fused | sed "s/^/[team32 $m q72] /" | tee -a $O/gemv.txt
done
timeout 300 python tools/team32_phase_probe.py 2>&1 | tee $O/phases.txt | tail -3
timeout 900 python bench.py --model qwen2-72b --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>$O/q72_err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen2-72b shapes, runner:', round(d['value'],2), 'tok/s', d['config'])" | tee $O/cfg4.txt
