set -u
R=/root/repo; M=/tmp/llama3-8b-q4k.bin
[ -s $M ] || python $R/tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out $M || exit 1
cd $R/oracle/_ref
REF_CHAT_FA=1 CLLM_HIP_STATS=1 ./ref_chat $M all 16 80 - 1 5 9 200 31 7 11 300 > /tmp/o.txt 2> /tmp/e.txt
grep "per graph" /tmp/e.txt | tail -2 | cut -c1-400; grep "^decode" /tmp/e.txt


