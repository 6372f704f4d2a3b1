#!/bin/bash
# round 5, call 40: BASELINE cfg5 / cfg4 through the unmodified host at FULL depth on the FINAL tree (the attention kernel and every soft_max changed since call 8): tokens/s + CPU-host-vs-module parity
O=gpurun_out/r5_40; mkdir -p $O
timeout 520 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 --dropin-cfg4 > $O/bench_line_dropin_cfg5_cfg4.json 2> $O/bench_err.txt
python - <<'PY' | tee -a gpurun_out/r5_40/summary.txt
import json
d=json.load(open('gpurun_out/r5_40/bench_line_dropin_cfg5_cfg4.json'))
print('value', round(d['value'],1), 'dropin', d.get('dropin_tok_s'))
for k in ('dropin_cfg5','dropin_cfg4_one_gpu'):
    x=d.get(k) or {}
    print(k, x.get('tok_s'), json.dumps(x.get('parity')), x.get('error'))
PY
