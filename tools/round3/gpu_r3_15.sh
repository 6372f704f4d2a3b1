#!/bin/bash
O=gpurun_out/r3o; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "vec_fused or packed_rows or gemv" 2>&1 | tail -4 | tee $O/pytest_ops.txt
cd /tmp && export TMPDIR=/tmp
for mode in exact fast; do
  rm -rf /tmp/prof_$mode
  CLLM_PREFILL=$mode timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -o p -- python /root/repo/tools/prefill_bench.py --layers 4 --reps 2 > /root/repo/$O/prefill_${mode}_4layers.log 2>&1
  f=$(find /tmp/prof_$mode -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -25 "$f" > /root/repo/$O/prefill_${mode}_kernel_stats.csv
  tail -2 /root/repo/$O/prefill_${mode}_4layers.log
done
cd /root/repo
# decode kernel stats of the bench (for profiles/r03_bench_kernel_stats.csv)
rm -rf /tmp/prof_dec; cd /tmp; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_dec -o p -- python /root/repo/bench.py --steps 48 --warmup 8 --no-cpu-baseline --no-pmc > /root/repo/$O/bench_prof.log 2>&1
f=$(find /tmp/prof_dec -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -30 "$f" > /root/repo/$O/bench_kernel_stats.csv
cd /root/repo; head -12 $O/bench_kernel_stats.csv | cut -c1-200
