#!/bin/bash
# round 6, call 15: the tensor-parallel device with launch-list replay (one captured graph per stream) and the lm_head's rows sharded; BASELINE cfg2 shapes through the host at 2 / 4 / 8
# virtual ranks on the one GPU (NOT a scaling number: the ranks share the GPU -- what it shows is what the replay removes)
O=gpurun_out/r6_15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel" -s 2>&1 | tail -25 | tee $O/pytest_tp.txt
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for n in 2 4 8; do
  for g in 0 1; do
    CLLM_HIP_TP=$n CLLM_HIP_TP_GRAPH=$g CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 96 - $IDS 2> $O/err_$n$g.txt | md5sum | tr '\n' ' ' | tee -a $O/tp_host.txt
    echo "CLLM_HIP_TP=$n replay=$g: $(grep 'decode:' $O/err_$n$g.txt)" | tee -a $O/tp_host.txt
  done
done
grep "tensor parallel" $O/err_81.txt | head -2 | cut -c1-250 | tee -a $O/tp_host.txt
grep "per graph over" $O/err_21.txt | tail -1 | tee -a $O/tp_host.txt
