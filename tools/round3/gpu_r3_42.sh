#!/bin/bash
# per-kernel times of the Q4_0 decode step (eager launches so that rocprofv3 names every kernel), team kernel on
O=$PWD/gpurun_out/r3x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pq -- python /root/repo/bench.py --wtype q4_0 --steps 64 --warmup 8 --no-graph --no-cpu-baseline --no-pmc --no-prefill > $O/bench_q4_0.log 2>&1
f=$(find /tmp/pq -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -16 $f > $O/q4_0_decode_kernel_stats.csv
python - <<'PY'
import csv
rows = list(csv.reader(open('/root/repo/gpurun_out/r3x/q4_0_decode_kernel_stats.csv')))
for r in rows[:12]: print(r[0][:70], r[1], r[3], r[4])
PY
