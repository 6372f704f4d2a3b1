#!/bin/bash
# round 4, call 15: k_mmf_exact with a 64 x 32 tile (three workgroups per CU) for the single-stage K.Q tiles: exactness + cfg3 prefill by forced tile
O=gpurun_out/r4_15; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "f16 or attn or attention or mul_mat_float" 2>&1 | tail -3 | tee $O/pytest.txt
CLLM_MMF_PM=1 timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "f16 or attn or attention or mul_mat_float" 2>&1 | tail -2 | tee -a $O/pytest.txt
for pm in 2 0 1 2 0; do
  CLLM_MMF_PM=$pm timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[CLLM_MMF_PM=$pm] /" | cut -c1-110 | tee -a $O/prefill.txt
done
timeout 900 python -m pytest tests/test_gpu_llama.py -m gpu -q -x -k "long_prompt" 2>&1 | tail -2 | tee -a $O/pytest.txt
