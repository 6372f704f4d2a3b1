#!/bin/bash
# round 5, call 37: bench.py's tensor-parallel set-up walked with ONE rank (CLLM_BENCH_TP_SELFTEST=1: RCCL communicator, one-shot and fused all-reduce self-checks, then the timed run)
O=gpurun_out/r5_37; mkdir -p $O
MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 CLLM_BENCH_TP_SELFTEST=1 timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-pmc --no-kernels --no-prefill > $O/line.json 2> $O/stderr.txt; echo "rc $?" | tee -a $O/summary.txt
grep -v "^$" $O/stderr.txt | tail -12 | cut -c1-220 | tee -a $O/summary.txt
python -c "import json; L=open('$O/line.json').read().splitlines(); print(len(L), 'stdout lines'); d=json.loads(L[0]); print(d['value'], d['config']['decode_allreduce'][:40])" | tee -a $O/summary.txt
