#!/bin/bash
# round 6, call 12: GPU time of the captured step alone vs the token period through the host (events around the graph launch; chain off so that the pair has fired when it is read)
O=gpurun_out/r6_12; mkdir -p $O
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for ch in 0 1; do echo "chain=$ch" >> $O/err.txt; CLLM_HIP_AHEAD_TIMING=1 CLLM_HIP_AHEAD_CHAIN=$ch CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2>> $O/err.txt > /dev/null; done
grep "chain=\|ahead timing\|decode:" $O/err.txt | tee $O/timing.txt
