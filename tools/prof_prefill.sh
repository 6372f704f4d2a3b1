#!/bin/bash
# kernel-trace of the cfg3 prefill (4 layers of the llama3-8b shapes, Q4_0, 4096 tokens): per-kernel totals
set -u
R=$PWD; O=$R/gpurun_out/${1:-prof_prefill}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python $R/tools/prefill_bench.py --layers ${LAYERS:-4} --reps 2 --wtype ${WTYPE:-q4_0} > $O/prefill_under_rocprof.txt 2>&1
cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) $O/prefill_kernel_stats.csv
head -25 $O/prefill_kernel_stats.csv | cut -c1-200
tail -3 $O/prefill_under_rocprof.txt
