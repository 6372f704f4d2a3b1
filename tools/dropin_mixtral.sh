#!/bin/bash
# BASELINE cfg5 through the UNMODIFIED reference host on our ggml module: Mixtral-8x7B shapes (8 experts, top 2, Q4_K, 26 GB), 16-token
# prompt (the reference feeds this architecture one token per graph) + 256 decoded tokens, one GPU.  Run on the GPU box.
set -u
R=/root/repo; M=/tmp/mixtral-8x7b-q4k.bin
[ -s $M ] || python $R/tools/make_ggmm.py --arch mixtral --config mixtral-8x7b --wtype q4_k --max-len 512 --fast --out $M || exit 1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd $R/oracle/_ref
s=$(date +%s); CLLM_HIP_STATS=1 ./ref_chat $M all ${THREADS:-16} ${N:-272} - $IDS > /tmp/mx_ids.txt 2> /tmp/mx_err.txt; echo "rc=$? wall=$(( $(date +%s) - s )) s"
grep "^decode:" /tmp/mx_err.txt; grep "per graph" /tmp/mx_err.txt | tail -1; grep "calls (" /tmp/mx_err.txt | tail -1; grep -i "error\|fail" /tmp/mx_err.txt | head -5
