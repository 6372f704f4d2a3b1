#!/bin/bash
# team kernel off / on in ONE call (box-to-box differences are a few per cent): decode tok/s at the default 512 steps, cfg4 runner
O=gpurun_out/r3u; mkdir -p $O; : > $O/onoff.txt
for t in q4_0 q8_0 q4_1; do for m in 0 1 0 1; do
  CLLM_GEMV_TEAM32=$m timeout 300 python bench.py --wtype $t --steps 512 --warmup 16 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t team32=$m decode', round(d['value'],1), 'tok/s')" | tee -a $O/onoff.txt
done; done
for m in 0 1; do
CLLM_GEMV_TEAM32=$m timeout 900 python bench.py --model qwen2-72b --steps 64 --warmup 8 --no-cpu-baseline --no-pmc 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('qwen2-72b shapes team32=$m, runner:', round(d['value'],2), 'tok/s')" | tee -a $O/onoff.txt
done
timeout 300 python tools/team32_phase_probe.py > $O/phases.txt 2>&1
