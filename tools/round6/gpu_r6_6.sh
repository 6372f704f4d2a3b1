#!/bin/bash
# round 6, call 6: (a) tensor parallel behind the ggml boundary (CLLM_HIP_TP = 2, 4, 8 virtual ranks through the unmodified host); (b) the chained decode-ahead (a second step
# queued behind the one the host waits for): the drop-in suite, then tok/s through the host at BASELINE cfg2 with the chain on / off
O=gpurun_out/r6_6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel_behind" -s 2>&1 | tail -25 | tee $O/pytest_tp.txt
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "not tensor_parallel_behind" 2>&1 | tail -8 | tee $O/pytest_dropin.txt
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for rep in 1 2; do
for ch in 0 1; do
  CLLM_HIP_AHEAD_CHAIN=$ch CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_chain$ch.txt | md5sum | tr '\n' ' ' | tee -a $O/dropin_ab.txt
  echo "CLLM_HIP_AHEAD_CHAIN=$ch: $(grep 'decode:' $O/err_chain$ch.txt)" | tee -a $O/dropin_ab.txt
done
done
grep "per graph over the last 64" $O/err_chain0.txt | tail -1 | tee -a $O/dropin_ab.txt
grep "per graph over the last 64" $O/err_chain1.txt | tail -1 | tee -a $O/dropin_ab.txt
grep "steps started ahead" $O/err_chain1.txt | tail -1 | tee -a $O/dropin_ab.txt
CLLM_HIP_TP=2 CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 64 - $IDS 2> $O/err_tp2.txt | md5sum | tee -a $O/dropin_ab.txt
echo "CLLM_HIP_TP=2 (two virtual ranks on the one GPU, eager launches): $(grep 'decode:' $O/err_tp2.txt)" | tee -a $O/dropin_ab.txt
grep "tensor parallel" $O/err_tp2.txt | head -3 | tee -a $O/dropin_ab.txt
