#!/bin/bash
# round 3, call 2: the exact-order prefill kernels -- parity tests, then cfg3 timing in both modes with per-kernel stats
O=gpurun_out/r3b; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "mul_mat" 2>&1 | tail -15 | tee $O/pytest_ops_mul_mat.txt
timeout 600 python -m pytest tests/test_gpu_fattn.py -q -x -k "attn_prefill" 2>&1 | tail -8 | tee $O/pytest_fattn.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x -k "long_prompt" 2>&1 | tail -15 | tee $O/pytest_llama_long.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -q -x -k "long_prompt" 2>&1 | tail -15 | tee $O/pytest_dropin_long.txt
for mode in exact fast; do
  CLLM_PREFILL=$mode timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep prefill | sed "s/^/[$mode] /" | tee -a $O/prefill_bench.txt
  CLLM_PREFILL=$mode timeout 300 python tools/prefill_bench.py --reps 3 --wtype q4_k 2>&1 | grep prefill | sed "s/^/[$mode] /" | tee -a $O/prefill_bench.txt
done
cd /tmp && export TMPDIR=/tmp
CLLM_PREFILL=exact timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_exact -o pf -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 2 --layers 4 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/prof_exact -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 $f | cut -c1-200 | tee $O/prefill_exact_kernel_stats_head.txt
