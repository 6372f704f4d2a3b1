#!/bin/bash
# round 4, call 32: BASELINE cfg5 (Mixtral-8x7B shapes) and cfg4 on one GPU (Qwen2-72B shapes) through the unmodified reference host on the module,
# by the bench line's own flags (--dropin-cfg5 / --dropin-cfg4)
O=gpurun_out/r4_32; mkdir -p $O
df -h /tmp | tail -1 | tee $O/disk.txt
timeout 1100 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 --dropin-cfg4 2>$O/stderr.txt > $O/bench_line_dropin_cfg5_cfg4.json
python - <<'PY' | tee $O/summary.txt
import json
d = json.load(open("gpurun_out/r4_32/bench_line_dropin_cfg5_cfg4.json"))
for k in ("dropin", "dropin_cfg5", "dropin_cfg4_one_gpu"):
    print(k, json.dumps(d.get(k))[:600])
PY
tail -5 $O/stderr.txt
