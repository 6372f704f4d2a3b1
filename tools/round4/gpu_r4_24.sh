#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 24: what bounds k_mmf_exact_vp?  timing-only variant builds: no P loads in the loop / no V loads / neither
O=gpurun_out/r4_24; mkdir -p $O
for v in vp_nocvt vp_nocvt_ld; do
  export CLLM_LIB=$GRAFT_REPO_ROOT/chatllm.cpp_amd/libchatllm_hip_$v.so
  cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$v -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof_$v -name "*kernel_stats.csv" | head -1); echo "variant=$v" | tee -a $O/vp_variants.txt; grep -E "k_mmf_exact_vp" "$f" | cut -c1-100 | tee -a $O/vp_variants.txt
  rm -rf $O/prof_$v
done
