#!/bin/bash
# round 5, call 20: BASELINE cfg5 (32 Mixtral blocks) through the host with the decode-ahead statistics: how many steps start ahead, how many hit
O=gpurun_out/r5_20; mkdir -p $O
M=/tmp/mixtral-8x7b-q4_k.bin
[ -s $M ] || python tools/make_ggmm.py --arch mixtral --config mixtral-8x7b --wtype q4_k --max-len 512 --fast --out $M 2>&1 | tail -1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd oracle/_ref
CLLM_HIP_STATS=1 CLLM_HIP_AHEAD_DEBUG=1 timeout 600 ./ref_chat $M all 16 272 - $IDS > /tmp/mx_ids.txt 2> /tmp/mx_err.txt; echo "rc=$?"
cd ../..
grep "^decode:" /tmp/mx_err.txt | tee -a $O/summary.txt
grep -c "replayed from the captured" /tmp/mx_err.txt | tee -a $O/summary.txt
grep "ahead:" /tmp/mx_err.txt | grep -v "ok=1" | head -20 | tee -a $O/summary.txt
grep "ahead: ok" /tmp/mx_err.txt | cut -c1-60 | sort | uniq -c | tee -a $O/summary.txt
grep "per graph\|replayed from a captured\|capture failed" /tmp/mx_err.txt | tee -a $O/summary.txt
