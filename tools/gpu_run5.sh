#!/bin/bash
mkdir -p gpurun_out/r2g
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_llama.py tests/test_golden.py -m gpu -q --timeout=300 -x > gpurun_out/r2g/pytest.log 2>&1; tail -4 gpurun_out/r2g/pytest.log | cut -c1-300
python tools/e2e_probe.py 2>&1 | cut -c1-200 | grep -v "decode:" > gpurun_out/r2g/e2e.log; grep -c "^OK" gpurun_out/r2g/e2e.log; grep "^DIFF\|TOTAL" gpurun_out/r2g/e2e.log
for n in 80 300 544 1000; do python tools/attn_phase_probe.py $n 2>/dev/null | grep -v "^\[rank"; done | tee gpurun_out/r2g/attn_phases.txt
python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-300
