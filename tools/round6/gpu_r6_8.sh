#!/bin/bash
# round 6, call 8: tensor parallel behind the boundary with chatllm's head (norm as a graph output + plain lm_head); the drop-in suite on the one-launch snapshot + arg-max;
# tok/s through the host: chain x one-launch prep A/B; token timeline again
O=gpurun_out/r6_8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "tensor_parallel_behind" -s 2>&1 | tail -25 | tee $O/pytest_tp.txt
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -k "not tensor_parallel_behind" 2>&1 | tail -8 | tee $O/pytest_dropin.txt
python tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 1024 --fast --out /tmp/l8.bin > $O/make.txt 2>&1
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
for rep in 1 2; do
for cfg in "0 0" "1 0" "1 1"; do
  set -- $cfg
  CLLM_HIP_AHEAD_CHAIN=$1 CLLM_HIP_AHEAD_ONE=$2 CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 272 - $IDS 2> $O/err_$1$2.txt | md5sum | tr '\n' ' ' | tee -a $O/dropin_ab.txt
  echo "chain=$1 one-launch-prep=$2: $(grep 'decode:' $O/err_$1$2.txt)" | tee -a $O/dropin_ab.txt
done
done
grep "steps started ahead" $O/err_11.txt | tail -1 | tee -a $O/dropin_ab.txt
CLLM_HIP_TP=2 CLLM_HIP_STATS=1 timeout 300 oracle/_ref/ref_chat /tmp/l8.bin all 4 64 - $IDS 2> $O/err_tp2.txt | md5sum | tee -a $O/dropin_ab.txt
echo "CLLM_HIP_TP=2 (two virtual ranks on the one GPU, eager launches): $(grep 'decode:' $O/err_tp2.txt)" | tee -a $O/dropin_ab.txt
grep "tensor parallel" $O/err_tp2.txt | head -3 | tee -a $O/dropin_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_1
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prof_1 -- $GRAFT_REPO_ROOT/oracle/_ref/ref_chat /tmp/l8.bin all 4 80 - $IDS > /dev/null 2> $GRAFT_REPO_ROOT/$O/prof_err.txt
python $GRAFT_REPO_ROOT/tools/round6/trace_token.py /tmp/prof_1 k_snapshot_argmax_set 2>&1 | tee -a $GRAFT_REPO_ROOT/$O/token_timeline.txt
