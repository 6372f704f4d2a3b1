cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pfa
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pfa -- python /root/repo/tools/fattn_bench.py 512 > /dev/null 2>&1
head -12 $(find /tmp/pfa -name "*kernel_stats.csv" | head -1) | cut -c1-150
