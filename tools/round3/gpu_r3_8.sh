#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "gemv or long_rows or vec_fused or packed_rows" 2>&1 | tail -6 | tee $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_llama.py -q -x 2>&1 | tail -4 | tee $O/pytest_llama.txt
timeout 600 python -m pytest tests/test_gpu_tp.py -q -x -k two_ranks 2>&1 | tail -3 | tee $O/pytest_tp.txt
timeout 300 python tools/gemv_bench.py --fused --types q4_0,q4_1,q8_0 --iters 64 2>&1 | grep fused | tee $O/gemv_fused_rows32.txt
CLLM_GEMV_ROWS32=0 timeout 300 python tools/gemv_bench.py --fused --types q4_0,q8_0 --iters 64 --shapes gate_up,down 2>&1 | grep fused | sed "s/^/[k_gemv_dec] /" | tee -a $O/gemv_fused_rows32.txt
for t in q4_0 q8_0 q4_1; do timeout 300 python bench.py --wtype $t --steps 128 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['value'],1), 'tok/s')" | tee -a $O/decode_other_types.txt; done
