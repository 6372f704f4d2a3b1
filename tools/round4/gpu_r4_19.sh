#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 19: K.Q as one workgroup per COLUMN tile walking its row tiles (k_mmf_exact_cols) against one workgroup per tile -- parity, then cfg3 prefill + kernel times
O=gpurun_out/r4_19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fattn.py tests/test_gpu_llama.py -m gpu -x -q -k "mul_mat_float or f16_exact or attn_prefill or prompt or prefill" 2>&1 | tail -4 | tee $O/tests.txt
for c in 0 2 1; do
  echo "CLLM_MMF_COLS=$c" | tee -a $O/prefill.txt
  CLLM_MMF_COLS=$c timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | tail -3 | tee -a $O/prefill.txt
done
for c in 0 2 1; do
  cd /tmp && export TMPDIR=/tmp && CLLM_MMF_COLS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$c -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/kernel_stats_cols$c.csv && echo "cols=$c" && head -8 $O/kernel_stats_cols$c.csv | cut -c1-150
  rm -rf $O/prof$c
done
