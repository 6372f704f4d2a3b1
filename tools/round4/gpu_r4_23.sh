#!/bin/bash
# (ran on the working tree of that moment: the k_mmf_exact_cols / k_mmf_exact_vp kernels, CLLM_MMF_COLS / CLLM_MMF_VP / CLLM_DEBUG_MMF and the MMF_T_* variant builds were removed afterwards;
#  results: profiles/r04_prompt_attention_kq_forms.txt.  What stayed: k_mmf_exact_kq (CLLM_MMF_KQ=0 turns it off) and the heads on grid x (CLLM_MMF_ZFIRST=0 restores the old order))
# round 4, call 23: V.P straight from global memory into the MFMA layout (k_mmf_exact_vp) against the LDS-staged tiles -- parity, then cfg3 prefill + kernel times
O=gpurun_out/r4_23; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fattn.py tests/test_gpu_llama.py -m gpu -x -q -k "mul_mat_float or f16_exact or attn_prefill or prompt or prefill" 2>&1 | tail -5 | tee $O/tests.txt
for vp in 0 1; do
  cd /tmp && export TMPDIR=/tmp && CLLM_MMF_VP=$vp timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 4 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof -name "*kernel_stats.csv" | head -1); echo "vp_form=$vp" | tee -a $O/vp.txt; grep -E "k_mmf_exact|soft_max" "$f" | cut -c1-110 | tee -a $O/vp.txt
  [ $vp = 1 ] && cp "$f" $O/prefill_exact_kernel_stats_4_layers.csv
  rm -rf $O/prof
done
for vp in 0 1; do
  echo "CLLM_MMF_VP=$vp" | tee -a $O/prefill.txt
  CLLM_MMF_VP=$vp timeout 600 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | tee -a $O/prefill.txt
done
