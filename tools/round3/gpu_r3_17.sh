#!/bin/bash
O=gpurun_out/r3q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for np in 2000 8000; do
rm -rf /tmp/prof_l; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_l -o p -- python /root/repo/bench.py --n-prompt $np --steps 32 --warmup 4 --no-cpu-baseline --no-pmc --no-graph > /root/repo/$O/long_$np.log 2>&1
f=$(find /tmp/prof_l -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -E "Name|attn_long|k_attn_dec|k_gemv_dec<12, 1, 1, 1" "$f" > /root/repo/$O/long_${np}_kernel_stats.csv
cat /root/repo/$O/long_${np}_kernel_stats.csv | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120
done
