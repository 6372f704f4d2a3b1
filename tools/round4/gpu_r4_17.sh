#!/bin/bash
# round 4, call 17: rocprofv3 kernel stats of 4 layers of the exact cfg3 prefill at the end of the round
O=gpurun_out/r4_17; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prefill_exact_kernel_stats_4_layers.csv && head -14 $O/prefill_exact_kernel_stats_4_layers.csv | cut -c1-170
rm -rf $O/prof
