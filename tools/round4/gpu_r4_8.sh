#!/bin/bash
# round 4, call 8: exact prefill attention with batched staging loads (k_mmf_exact) and batched row loads (k_soft_max_causal_reg): exactness tests + cfg3 prefill time
O=gpurun_out/r4_8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "f16 or soft_max or attn or attention or mul_mat_float" 2>&1 | tail -4 | tee $O/pytest.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x 2>&1 | tail -3 | tee -a $O/pytest.txt
timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | grep -v "^\[rank" | tail -2 | tee $O/prefill.txt
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" | cut -c1-200 | tee $O/prefill_kernel_stats_head.txt
