cd /tmp && export TMPDIR=/tmp
for d in 8 4; do
  rm -rf /tmp/pp; CLLM_MMVQ_DEPTH=$d rocprofv3 --kernel-trace --output-format csv -d /tmp/pp -- python /root/repo/bench.py --steps 48 --warmup 8 --no-cpu-baseline > /tmp/pp.log 2>&1
  echo "== depth $d"; grep -o '"value": [0-9.]*' /tmp/pp.log
  python /root/repo/tools/trace_token.py $(find /tmp/pp -name "*kernel_trace.csv" | head -1) 7
done
