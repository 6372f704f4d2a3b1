#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q -x --durations=25 2>&1 | tail -45 | tee $O/pytest_gpu_full.txt
timeout 900 python bench.py > $O/bench_line.json 2> $O/bench_stderr.txt; tail -5 $O/bench_stderr.txt; cat $O/bench_line.json | cut -c1-1500
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
