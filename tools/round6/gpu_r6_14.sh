#!/bin/bash
# round 6, call 14: the evidence run -- whole GPU suite, the bench line exactly as the driver runs it (timed), rocprofv3 kernel stats of the bench
O=gpurun_out/r6_14; mkdir -p $O
timeout 1800 python -m pytest tests/ -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu_full.txt; tail -3 $O/pytest_gpu_full.txt
t0=$(date +%s)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_stderr.txt > $O/bench_line.json; echo "bench rc $? wall $(( $(date +%s) - t0 )) s" | tee $O/bench_wall.txt; cut -c1-300 $O/bench_line.json
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 56 --warmup 8 --no-cpu-baseline --no-pmc --no-kernels --no-prefill --no-other-types --no-graph > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/bench_kernel_stats.csv && head -8 $O/bench_kernel_stats.csv | cut -c1-160
rm -rf $O/prof
