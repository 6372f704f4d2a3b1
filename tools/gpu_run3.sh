#!/bin/bash
mkdir -p gpurun_out/r2d
for m in 1 2 8 4; do CLLM_GEMV_ROWS=$m timeout 300 python tools/exact_probe.py > gpurun_out/r2d/exact_probe_rows$m.log 2>&1; echo "rows=$m: $(grep -c '^OK' gpurun_out/r2d/exact_probe_rows$m.log) ok; $(grep -c '^DIFF' gpurun_out/r2d/exact_probe_rows$m.log) diff; $(tail -1 gpurun_out/r2d/exact_probe_rows$m.log)"; grep "^DIFF" gpurun_out/r2d/exact_probe_rows$m.log | head -8; done
for m in 0 1 8 4; do echo "== rows=$m"; CLLM_GEMV_ROWS=$m timeout 300 python tools/gemv_bench.py --fused --types q4_k --iters 128 2>&1 | grep fused; done | tee gpurun_out/r2d/gemv_bench.log
