#!/bin/bash
# round 6, call 5: the driver's bench command on the new bench.py (other_types legs, CPU baseline sweep + AVX-512 build), timed; then the whole GPU suite on the committed tree
O=gpurun_out/r6_5; mkdir -p $O
t0=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_err.txt
echo "bench rc $? wall $(( $(date +%s) - t0 )) s" | tee $O/bench_wall.txt
tail -c 1500 $O/bench_err.txt
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee $O/pytest_gpu.txt
