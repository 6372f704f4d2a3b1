#!/bin/bash
# round 3, call 6: persistent decode launch -- where a phase's time goes (stamps) and how it moves with the run-ahead depth; two-rank TP test
O=gpurun_out/r3f; mkdir -p $O
timeout 300 python tools/persist_phase_probe.py 2>&1 | tail -8 | tee $O/persist_phases.txt
for v in pgp1 "" pgp3 pgp4 pgns; do
  lib=$PWD/chatllm.cpp_amd/libchatllm_hip${v:+_$v}.so
  CLLM_LIB=$lib timeout 300 python bench.py --steps 64 --warmup 16 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[${v:-pgp2}]', round(d['value'],1), 'tok/s', round(d['ms_per_step']*1000/32,2), 'us/layer+')" | tee -a $O/persist_variants.txt
done
timeout 600 python -m pytest tests/test_gpu_tp.py -q -x 2>&1 | tail -8 | tee $O/pytest_tp.txt
