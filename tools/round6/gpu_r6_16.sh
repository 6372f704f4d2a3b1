#!/bin/bash
# round 6, call 16: BASELINE cfg5 (Mixtral-8x7B shapes) and cfg4 (Qwen2-72B shapes, one GPU) through the unmodified host on the final tree, with the full-depth parity check
O=gpurun_out/r6_16; mkdir -p $O
timeout 3000 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --no-other-types --dropin-cfg5 --dropin-cfg4 2>$O/bench_err.txt > $O/bench_line.json
python - <<'P' | tee $O/summary.txt
import json
d = json.loads(open('gpurun_out/r6_16/bench_line.json').read().strip().splitlines()[-1])
for k in ('dropin', 'dropin_cfg5', 'dropin_cfg4'):
    v = d.get(k)
    print(k, json.dumps(v)[:900] if v else None)
P
