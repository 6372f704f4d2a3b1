#!/bin/bash
# Decode throughput of the UNMODIFIED reference host (oracle/_ref/ref_chat) on our ggml backend module (-ngl all) vs its own CPU
# backend, Llama-3-8B shapes Q4_K synthetic model in GGMM format.  Run on the GPU box: bash tools/dropin_bench.sh
set -u
R=/root/repo; M=/tmp/llama3-8b-q4k.bin
[ -s $M ] || python $R/tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out $M || exit 1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd $R/oracle/_ref
for ngl in ${NGL:-all}; do
  for n in ${NS:-144 400 144 400}; do
    s=$(date +%s%N); ./ref_chat $M $ngl ${THREADS:-16} $n - $IDS > /tmp/ids_$n.txt 2>/tmp/err_$n.txt; rc=$?; e=$(date +%s%N)
    echo "ngl=$ngl n_decode=$n wall_ms=$(( (e - s) / 1000000 )) rc=$rc"; grep "^decode:" /tmp/err_$n.txt | cut -c1-200
  done
done
