#!/bin/bash
# round 5, call 10: the sparse-MoE router folded into the experts' gate / up launch (5 launches per Mixtral block instead of 6)
O=gpurun_out/r5_10; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "moe or mul_mat_id or fused" 2>&1 | tail -4 | tee -a $O/summary.txt
timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -x -k "mixtral" 2>&1 | tail -4 | tee -a $O/summary.txt
CLLM_HIP_MOE_FOLD=0 timeout 1500 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 --no-full-depth-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold=0', json.dumps({k: d['dropin_cfg5'].get(k) for k in ('tok_s','calls_per_token','breakdown_us')}))" | tee -a $O/summary.txt
CLLM_HIP_MOE_FOLD=1 timeout 1500 python bench.py --steps 20 --warmup 5 --no-pmc --no-kernels --no-prefill --dropin-cfg5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fold=1', json.dumps({k: d['dropin_cfg5'].get(k) for k in ('tok_s','calls_per_token','breakdown_us','parity')}))" | tee -a $O/summary.txt
