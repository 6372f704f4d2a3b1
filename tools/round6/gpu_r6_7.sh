#!/bin/bash
# round 6, call 7: (a) tensor parallel behind the boundary, test shapes fixed (head_dim = hidden / n_head in the reference's loader); (b) where a token's time goes on the GPU
# through the unmodified host: rocprofv3 kernel + copy trace of ref_chat -ngl all at BASELINE cfg2 shapes, one steady-state token as a timeline
O=gpurun_out/r6_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel_behind" -s 2>&1 | tail -25 | tee $O/pytest_tp.txt
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd /tmp && export TMPDIR=/tmp
for ch in 1 0; do
  rm -rf /tmp/prof_$ch
  CLLM_HIP_AHEAD_CHAIN=$ch timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_$ch -- $GRAFT_REPO_ROOT/oracle/_ref/ref_chat /tmp/l8.bin all 4 80 - $IDS > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err_$ch.txt
  echo "== CLLM_HIP_AHEAD_CHAIN=$ch ==" | tee -a $GRAFT_REPO_ROOT/$O/token_timeline.txt
  python $GRAFT_REPO_ROOT/tools/round6/trace_token.py /tmp/prof_$ch k_argmax_publish_set 2>&1 | tee -a $GRAFT_REPO_ROOT/$O/token_timeline.txt
done
