#!/bin/bash
mkdir -p gpurun_out; O=gpurun_out/r3y.txt; : > $O
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "dense_f16" 2>&1 | tail -3 >> $O
for tile in ${TILES:-128 256}; do
  CLLM_PREFILL=f16 CLLM_MMD_TILE=$tile timeout 300 python tools/gemv_bench.py --types ${TYPES:-q4_0,q4_k} --cols 4096 --iters 8 --shapes qkv,o,gate_up,down 2>&1 | grep -E "K=" | sed "s/^/[tile $tile] /" >> $O
done
cat $O
