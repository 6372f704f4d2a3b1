#!/bin/bash
# round 5, call 8: BASELINE cfg5 / cfg4 through the unmodified host at FULL depth (32 / 80 layers): tokens/s + CPU-host-vs-module parity of ids and logit words; quantizer tests of the leaner 4-per-lane form
O=gpurun_out/r5_8; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "quantize or fused or norm_prologues or quant_gemv or moe or mul_mat_id" 2>&1 | tail -3 | tee -a $O/summary.txt
timeout 3000 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 --dropin-cfg4 > $O/bench_line_dropin_cfg5_cfg4.json 2> $O/bench_err.txt
python - <<'PY' | tee -a gpurun_out/r5_8/summary.txt
import json
d=json.load(open('gpurun_out/r5_8/bench_line_dropin_cfg5_cfg4.json'))
print('value', round(d['value'],1), 'decode_512', d.get('decode_512',{}).get('value'), 'dropin', d.get('dropin_tok_s'), 'cpu', d.get('cpu_baseline',{}).get('value'))
for k in ('dropin_cfg5','dropin_cfg4_one_gpu'):
    print(k, json.dumps(d.get(k)))
PY
tail -5 $O/bench_err.txt
