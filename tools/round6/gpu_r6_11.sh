#!/bin/bash
# round 6, call 11: the RUNNER's token on the same tracer (is the gap pattern of the host path the tracer's or the path's?)
O=gpurun_out/r6_11; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_r
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_r -- python $GRAFT_REPO_ROOT/bench.py --steps 64 --warmup 8 --no-pmc --no-kernels --no-prefill --no-cpu-baseline --no-other-types > $GRAFT_REPO_ROOT/$O/bench_out.txt 2> $GRAFT_REPO_ROOT/$O/prof_err.txt
python $GRAFT_REPO_ROOT/tools/round6/trace_token.py /tmp/prof_r k_argmax_final_next 2>&1 | tee $GRAFT_REPO_ROOT/$O/token_timeline_runner.txt
