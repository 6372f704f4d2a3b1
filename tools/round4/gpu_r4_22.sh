#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 22: the heads on grid x (every head's longest causal tiles dispatched first, an XCD keeps one K/V head) against the heads on grid z -- the three K.Q forms and V.P
O=gpurun_out/r4_22; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fattn.py tests/test_gpu_llama.py -m gpu -x -q -k "mul_mat_float or f16_exact or attn_prefill or prompt or prefill" 2>&1 | tail -3 | tee $O/tests.txt
for zf in 0 1; do for c in 0 2 1; do
  cd /tmp && export TMPDIR=/tmp && CLLM_MMF_ZFIRST=$zf CLLM_MMF_COLS=$c timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "zfirst=$zf kq_form=$c" | tee -a $O/grid_order.txt; grep -E "k_mmf_exact" "$f" | cut -c1-110 | tee -a $O/grid_order.txt
  rm -rf $O/prof
done; done
for zf in 0 1; do
  echo "CLLM_MMF_ZFIRST=$zf" | tee -a $O/prefill.txt
  CLLM_MMF_ZFIRST=$zf timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | tee -a $O/prefill.txt
done
