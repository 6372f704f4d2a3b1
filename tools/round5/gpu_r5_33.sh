#!/bin/bash
# round 5, call 33: BASELINE cfg5 through the unmodified host on the final tree (the attention rework is in every block)
O=gpurun_out/r5_33; mkdir -p $O
timeout 1500 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 --no-full-depth-parity 2>/dev/null > $O/bench_line_cfg5.json
python -c "import sys,json; d=json.load(open('$O/bench_line_cfg5.json')); print(json.dumps({k: d['dropin_cfg5'].get(k) for k in ('tok_s','calls_per_token','breakdown_us')})); print('llama dropin', d.get('dropin_tok_s'), 'value', d['value'])" | tee -a $O/summary.txt
