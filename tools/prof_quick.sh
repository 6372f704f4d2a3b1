#!/bin/bash
# quick kernel-trace of the bench command (decode steps at ~n_ctx 300): per-kernel averages + one decode step's launch sequence
set -u
R=$PWD; O=$R/gpurun_out/${1:-prof_quick}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --n-prompt ${NPROMPT:-256} --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_token.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 14 > $O/decode_step_trace.txt
cat $O/decode_step_trace.txt
