#!/bin/bash
# round 4, call 12: the module starts the next decode step BEFORE the logits copy (CLLM_HIP_AHEAD_LATE=1: round 3's order): drop-in tok/s A/B + the drop-in suite
O=gpurun_out/r4_12; mkdir -p $O
M=/tmp/llama3-8b-q4k.bin
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out $M > /dev/null 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for rep in 1 2; do
  for late in 1 0; do
    if [ $late = 1 ]; then export CLLM_HIP_AHEAD_LATE=1; else unset CLLM_HIP_AHEAD_LATE; fi
    CLLM_HIP_STATS=1 oracle/_ref/ref_chat $M all 16 272 - $IDS 2> $O/err.txt | tail -c 60 | tr '\n' ' '; echo "late=$late: $(grep '^decode:' $O/err.txt) | $(grep 'per graph' $O/err.txt | tail -1 | cut -c1-200)" | tee -a $O/dropin_ab.txt
  done
done
unset CLLM_HIP_AHEAD_LATE
timeout 1500 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x 2>&1 | tail -4 | tee $O/pytest_dropin.txt
