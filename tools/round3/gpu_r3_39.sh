#!/bin/bash
O=$PWD/gpurun_out/r3v; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
rm -rf /tmp/q72; CLLM_GEMV_TEAM32=$m timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q72 -- python /root/repo/tools/q72_kernel_mix.py > $O/run_$m.log 2>&1
f=$(find /tmp/q72 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && python -c "import csv,sys; [print(r[0][:60].replace(chr(10),\" \"), r[1], r[3], r[4]) for r in list(csv.reader(open(sys.argv[1])))[:12]]" $f > $O/kernel_stats_$m.csv
done
tail -2 $O/run_1.log; cat $O/kernel_stats_0.csv; cat $O/kernel_stats_1.csv
