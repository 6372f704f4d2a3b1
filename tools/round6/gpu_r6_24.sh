#!/bin/bash
# round 6, call 24: BASELINE cfg4 (Qwen2-72B shapes, Q4_K with the Q8_0 down projection, 80 layers) through the unmodified host: one device against CLLM_HIP_TP=8 with the eight
# ranks as VIRTUAL ranks on the one GPU (their launches run one after the other).  Not a scaling measurement -- but the serialized time of the eight ranks / 8 is what one rank of a
# real 8-GPU node would spend on its share (plus the granules' flight over xGMI): a projection with a stated model.
O=gpurun_out/r6_24; mkdir -p $O
M=/tmp/qwen2-72b-q4_k.bin
python tools/make_ggmm.py --arch qwen2 --config qwen2-72b --wtype q4_k --max-len 512 --fast --out $M > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
CLLM_HIP_STATS=1 timeout 900 oracle/_ref/ref_chat $M all 8 80 - $IDS 2> $O/err_1.txt | md5sum | tr '\n' ' ' | tee -a $O/tp72.txt; echo "one device: $(grep 'decode:' $O/err_1.txt)" | tee -a $O/tp72.txt
for n in 2 8; do
  CLLM_HIP_TP=$n CLLM_HIP_STATS=1 timeout 1200 oracle/_ref/ref_chat $M all 8 80 - $IDS 2> $O/err_$n.txt | md5sum | tr '\n' ' ' | tee -a $O/tp72.txt
  echo "CLLM_HIP_TP=$n (virtual ranks on the one GPU): $(grep 'decode:' $O/err_$n.txt)" | tee -a $O/tp72.txt
  grep "tensor parallel: " $O/err_$n.txt | head -1 | cut -c1-250 | tee -a $O/tp72.txt
done
rm -f $M
