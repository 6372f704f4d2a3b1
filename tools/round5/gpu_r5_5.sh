#!/bin/bash
# round 5, call 5: decode attention with 512 K rows requested up front + the LDS running maximum; the new bench line fields (decode_512, layer_split)
O=gpurun_out/r5_5; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py -m gpu -q -x -k "attn or attention or rope_kv or llama or decode or fused" 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 300 python tools/attn_phase_probe.py 2>&1 | tee $O/attn_stamps.txt | tail -24
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --no-prefill > $O/bench_line.json 2> $O/bench_err.txt
python - <<'PY' | tee -a gpurun_out/r5_5/summary.txt
import json
d=json.load(open('gpurun_out/r5_5/bench_line.json'))
print('value', round(d['value'],1), 'decode_512', d.get('decode_512'), 'dtype', d.get('dtype'))
for r in d.get('kernels', {}).get('launches', d.get('kernels', [])) if isinstance(d.get('kernels'), dict) else d.get('kernels', []):
    print(r)
print(json.dumps(d.get('layer_split'), indent=1))
print(d.get('roofline'))
PY
tail -5 $O/bench_err.txt
