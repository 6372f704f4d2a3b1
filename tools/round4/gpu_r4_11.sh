#!/bin/bash
# round 4, call 11: per-kernel times of the fast prefill mode, round 3's library vs this round's (where do 17 ms go?), after the fast serial fallback
O=gpurun_out/r4_11; mkdir -p $O
for lib in _r03 ""; do
  CLLM_PREFILL=fast CLLM_LIB=$PWD/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[lib${lib:-_r04} fast] /" | cut -c1-120 | tee -a $O/prefill_ab.txt
  cd /tmp && export TMPDIR=/tmp && CLLM_PREFILL=fast CLLM_LIB=$GRAFT_REPO_ROOT/chatllm.cpp_amd/libchatllm_hip$lib.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$lib -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 2 --layers 8 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof$lib -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/fast_kernel_stats${lib:-_r04}.csv && head -14 $O/fast_kernel_stats${lib:-_r04}.csv | cut -c1-150
  rm -rf $O/prof$lib
done
CLLM_PREFILL=exact timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | grep "^prefill" | sed "s/^/[lib_r04 exact] /" | cut -c1-120 | tee -a $O/prefill_ab.txt
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "rms or norm_prologues" 2>&1 | tail -3
