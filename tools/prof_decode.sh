#!/bin/bash
# rocprofv3 kernel trace of the default bench (on the GPU box): per-kernel table + the launch sequence of one decode step
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python /root/repo/bench.py --steps ${STEPS:-48} --warmup 8 --no-cpu-baseline > /tmp/pp.log 2>&1
python /root/repo/tools/trace_token.py $(find /tmp/pp -name "*kernel_trace.csv" | head -1) ${ROWS:-10}
mkdir -p /root/repo/gpurun_out/prof_last && cp $(find /tmp/pp -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/prof_last/kernel_stats.csv
