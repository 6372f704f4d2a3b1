#!/bin/bash
# round 5, call 32: the evidence run at the end of the round -- whole GPU suite, the bench line (default and the driver's --steps 20 --warmup 5), rocprofv3 kernel stats of the bench
O=gpurun_out/r5_32; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu_full.txt; tail -3 $O/pytest_gpu_full.txt
timeout 1200 python bench.py 2>$O/bench_stderr.txt > $O/bench_line.json; cut -c1-300 $O/bench_line.json
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-kernels 2>/dev/null > $O/bench_line_steps20.json; cut -c1-200 $O/bench_line_steps20.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 56 --warmup 8 --no-cpu-baseline --no-pmc --no-kernels --no-prefill --no-graph > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
