#!/bin/bash
# round 6, call 10: the runner and the unmodified host at the SAME contexts (16-token prompt, 16 untimed steps, 256 timed: context 32 -> 288), back to back on one box
O=gpurun_out/r6_10; mkdir -p $O
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for rep in 1 2; do
  timeout 400 python bench.py --steps 256 --warmup 16 --no-pmc --no-kernels --no-prefill --no-cpu-baseline --no-other-types 2>$O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('runner (bench.py --steps 256 --warmup 16): %.1f tok/s  %.4f ms/step  n_ctx_end %s' % (d['value'], d['ms_per_step'], d['config'].get('n_ctx_end')))" | tee -a $O/same_context.txt
  CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_host.txt > /dev/null
  echo "host   (ref_chat -ngl all, 272 steps, 16 untimed): $(grep 'decode:' $O/err_host.txt)" | tee -a $O/same_context.txt
done
grep "per graph over the last 64" $O/err_host.txt | tail -1 | tee -a $O/same_context.txt
