#!/bin/bash
# round 5, call 6: decode attention (next pass's K rows requested before this pass's math; per-wave max + one LDS atomic), Q16 prologue on the MoE down + combine launch
O=gpurun_out/r5_7; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py -m gpu -q -x -k "attn or attention or rope_kv or llama or decode or fused or moe or mul_mat_id" 2>&1 | tail -4 | tee -a $O/summary.txt
for n in 80 128 300 544; do timeout 300 python tools/attn_phase_probe.py $n 2>&1 | tail -6 | tee -a $O/attn_stamps.txt; done
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels --no-prefill"
$B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
$B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('steps20  %.1f tok/s  decode_512 %.1f  tail %s' % (d['value'], d['decode_512']['value'], d['greedy_tail']))" | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k "mixtral" 2>&1 | tail -6 | tee -a $O/summary.txt
