#!/bin/bash
# round 4, call 31: the evidence run at the end of the round (after the attention grid order, k_mmf_exact_kq and the RMS fallback change) -- whole GPU suite, the bench line (default and the driver's --steps 20 --warmup 5), rocprofv3 kernel stats of the bench
O=gpurun_out/r4_31; mkdir -p $O
timeout 1500 python -m pytest tests/ -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu_full.txt; tail -3 $O/pytest_gpu_full.txt
timeout 900 python bench.py 2>$O/bench_stderr.txt > $O/bench_line.json; cut -c1-300 $O/bench_line.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/bench_line_steps20.json; cut -c1-200 $O/bench_line_steps20.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 56 --warmup 8 --no-cpu-baseline --no-pmc --no-kernels --no-graph > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof2 -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof2 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/prefill_exact_kernel_stats_4_layers.csv && head -6 $O/prefill_exact_kernel_stats_4_layers.csv | cut -c1-140
rm -rf $O/prof2
