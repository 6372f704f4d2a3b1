#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 20: what bounds k_mmf_exact_cols?  (1) no score stores at all, (2) non-temporal stores, against the plain stores
O=gpurun_out/r4_20; mkdir -p $O
for dbg in 0 1 2; do
  cd /tmp && export TMPDIR=/tmp && CLLM_DEBUG_MMF=$dbg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$dbg -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof$dbg -name "*kernel_stats.csv" | head -1); echo "dbg=$dbg" | tee -a $O/stores.txt; grep -E "k_mmf_exact|soft_max" "$f" | cut -c1-120 | tee -a $O/stores.txt
  rm -rf $O/prof$dbg
done
