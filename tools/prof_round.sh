#!/bin/bash
# Evidence for profiles/ (run on the GPU box through gpurun): kernel-trace stats of the default bench command, the bench line itself,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel-trace only) on the dominant kernel.
set -u
R=/root/repo; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_token.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 12 > $O/decode_step_trace.txt
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8 > $O/pmc_fetch.log 2>&1
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3 -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8 > $O/pmc_write.log 2>&1
cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
python $R/tools/gemv_bench.py --fused --types q4_k > $O/gemv_fused.txt 2>&1
python $R/tools/gemv_bench.py > $O/gemv_plain.txt 2>&1
python $R/tools/gemv_phase_probe.py > $O/gemv_phases.txt 2>&1
tail -3 $O/bench.json; cat $O/decode_step_trace.txt | head -16
