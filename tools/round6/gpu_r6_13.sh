#!/bin/bash
# round 6, call 13: the chained decode-ahead with EVERY synchronize() of a token kept off the stream (the host calls it more than once per token: the second call used to wait
# for the step running ahead); runner and host at the same contexts, chain on / off, GPU-clock timing
O=gpurun_out/r6_13; mkdir -p $O
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for rep in 1 2; do
for ch in 0 1; do
  CLLM_HIP_AHEAD_CHAIN=$ch CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_$ch.txt | md5sum | tr '\n' ' ' | tee -a $O/ab.txt
  echo "chain=$ch: $(grep 'decode:' $O/err_$ch.txt)" | tee -a $O/ab.txt
done
done
grep "per graph over the last 64" $O/err_0.txt | tail -1 | tee -a $O/ab.txt
grep "per graph over the last 64" $O/err_1.txt | tail -1 | tee -a $O/ab.txt
CLLM_HIP_AHEAD_TIMING=1 CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_t.txt > /dev/null; grep "ahead timing" $O/err_t.txt | tee -a $O/ab.txt
timeout 400 python bench.py --steps 256 --warmup 16 --no-pmc --no-kernels --no-prefill --no-cpu-baseline --no-other-types 2>$O/bench_err.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('runner (bench.py --steps 256 --warmup 16): %.1f tok/s  %.4f ms/step' % (d['value'], d['ms_per_step']))" | tee -a $O/ab.txt
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_dropin.txt
