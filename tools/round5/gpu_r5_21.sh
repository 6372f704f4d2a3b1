#!/bin/bash
# round 5, call 21: rocprofv3 kernel stats of the Mixtral drop-in (8 real-shape blocks, 16-token prompt + 128 decoded tokens; graph replay off so that every launch is traced by name)
O=gpurun_out/r5_21; mkdir -p $O
M=/tmp/mx8.bin
python tools/make_ggmm.py --arch mixtral --config mixtral-8x7b --wtype q4_k --max-len 512 --fast --layers 8 --out $M 2>&1 | tail -1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CLLM_HIP_GRAPH=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -- $R/oracle/_ref/ref_chat $M all 16 128 - $IDS > /tmp/mx_ids.txt 2> /tmp/mx_err.txt; echo "rc=$?"
cd $R
grep "^decode:" /tmp/mx_err.txt | tee -a $O/summary.txt
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/mixtral_8_blocks_kernel_stats.csv && head -14 $O/mixtral_8_blocks_kernel_stats.csv | cut -c1-200 | tee -a $O/summary.txt
rm -rf $O/prof
