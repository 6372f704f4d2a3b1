#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 26: the register-direct K.Q / V.P kernels on the UNIFORM problem (every tile / every position computed: timing only) -- is the causal run's distance
# from the matrix-core time the kernels' steady-state rate, or the causal shape (work per workgroup from 1 to 128 steps)?
O=gpurun_out/r4_26; mkdir -p $O
for dbg in 0 3 4; do
  cd /tmp && export TMPDIR=/tmp && CLLM_DEBUG_MMF=$dbg timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "dbg=$dbg (3: V.P over every position, 4: K.Q over every row)" | tee -a $O/uniform.txt; grep -E "k_mmf_exact" "$f" | cut -c1-100 | tee -a $O/uniform.txt
  rm -rf $O/prof
done
