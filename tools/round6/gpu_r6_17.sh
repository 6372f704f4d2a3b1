#!/bin/bash
# round 6, call 17: where the other two weight types of the north star lose their time: the per-launch table and the in-kernel layer split of the decode step for Q4_0 and Q8_0
O=gpurun_out/r6_17; mkdir -p $O
for wt in q4_0 q8_0; do
  timeout 600 python bench.py --wtype $wt --steps 20 --warmup 5 --no-pmc --no-prefill --no-cpu-baseline --no-other-types 2>$O/err_$wt.txt > $O/line_$wt.json
  python - $wt <<'P' | tee -a $O/summary.txt
import json, sys
wt = sys.argv[1]
d = json.loads(open(f'gpurun_out/r6_17/line_{wt}.json').read().strip().splitlines()[-1])
print(wt, 'value', round(d['value'], 1), 'ms/step', round(d['ms_per_step'], 4), 'model_frac', round(d['model_hbm_frac'], 3))
for k in d.get('kernels', {}).get('per_launch', []): print('   %-80s %7.2f us  %6.1f GB/s  %.3f' % (k['name'][:80], k['us'], k['gbs'], k['frac']))
print('   layer_split', json.dumps(d.get('layer_split'))[:1200])
P
done
