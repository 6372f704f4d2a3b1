#!/bin/bash
# BASELINE cfg4 on ONE GPU through the UNMODIFIED reference host on our ggml module: Qwen2-72B shapes (Q4_K, down_proj Q8_0, q/k/v biases,
# NEOX RoPE; 50 GB), 16-token prompt + 128 decoded tokens.  Run on the GPU box.
set -u
R=/root/repo; M=/tmp/qwen2-72b-q4k.bin
[ -s $M ] || python $R/tools/make_ggmm.py --arch qwen2 --config qwen2-72b --wtype q4_k --max-len 512 --fast --out $M || exit 1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd $R/oracle/_ref
s=$(date +%s); CLLM_HIP_STATS=1 ./ref_chat $M all ${THREADS:-16} ${N:-144} - $IDS > /tmp/qw_ids.txt 2> /tmp/qw_err.txt; echo "rc=$? wall=$(( $(date +%s) - s )) s"
grep "^decode:" /tmp/qw_err.txt; grep "per graph" /tmp/qw_err.txt | tail -1; grep "calls (" /tmp/qw_err.txt | tail -1; grep -i "error\|fail" /tmp/qw_err.txt | head -5
