#!/bin/bash
# round 6, call 22: decode-ahead on the tensor-parallel device (the next sharded step started ahead of the host): tests, the drop-in suite, 8B shapes at 2 virtual ranks on / off
O=gpurun_out/r6_22; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel" -s 2>&1 | grep -E "passed|failed|rror|ranks" | tail -25 | tee $O/pytest_tp.txt
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "not tensor_parallel and not free_order" 2>&1 | grep -E "passed|failed|rror" | tee $O/pytest_dropin.txt
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for n in 2 8; do for a in 0 1; do
  CLLM_HIP_TP=$n CLLM_HIP_AHEAD=$a CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 96 - $IDS 2> $O/err_$n$a.txt | md5sum | tr '\n' ' ' | tee -a $O/tp_host.txt
  echo "CLLM_HIP_TP=$n decode-ahead=$a: $(grep 'decode:' $O/err_$n$a.txt)  steps started ahead: $(grep -c 'started ahead of the host' $O/err_$n$a.txt)" | tee -a $O/tp_host.txt
done; done
CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_single.txt | md5sum | tr '\n' ' ' | tee -a $O/tp_host.txt; echo "single device: $(grep 'decode:' $O/err_single.txt)" | tee -a $O/tp_host.txt
