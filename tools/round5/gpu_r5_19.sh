#!/bin/bash
# round 5, call 19: why the Mixtral launch list is not replayed from a captured graph (no decode-ahead then): 4 real-shape blocks through the host with the module's debug switches
O=gpurun_out/r5_19; mkdir -p $O
M=/tmp/mx4.bin
python tools/make_ggmm.py --arch mixtral --config mixtral-8x7b --wtype q4_k --max-len 512 --fast --layers 4 --out $M 2>&1 | tail -2
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd oracle/_ref
CLLM_HIP_STATS=1 CLLM_HIP_SIG_DEBUG=1 CLLM_HIP_AHEAD_DEBUG=1 timeout 300 ./ref_chat $M all 16 96 - $IDS > /tmp/mx_ids.txt 2> /tmp/mx_err.txt; echo "rc=$?"
cd ../..
grep "^decode:" /tmp/mx_err.txt | tee -a $O/summary.txt
grep -c "replayed from the captured" /tmp/mx_err.txt | tee -a $O/summary.txt
grep "launch list differs" /tmp/mx_err.txt | sort | uniq -c | sort -rn | head -12 | tee -a $O/summary.txt
grep "ahead:" /tmp/mx_err.txt | grep -v "ok=1" | head -30 | tee -a $O/summary.txt
grep "per graph\|replayed from a captured\|capture failed" /tmp/mx_err.txt | tail -4 | tee -a $O/summary.txt
head -c 20000 /tmp/mx_err.txt > $O/err_head.txt
