#!/bin/bash
mkdir -p gpurun_out/r2b
python tools/exact_probe.py > gpurun_out/r2b/exact_probe.log 2>&1
python tools/e2e_probe.py > gpurun_out/r2b/e2e_probe.log 2>&1
timeout 900 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/r2b/pytest_all.log 2>&1
grep -c "^OK" gpurun_out/r2b/exact_probe.log; grep -v "^OK" gpurun_out/r2b/exact_probe.log | head -20
cut -c1-330 gpurun_out/r2b/e2e_probe.log
tail -30 gpurun_out/r2b/pytest_all.log
