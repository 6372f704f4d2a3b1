set -u
R=/root/repo; O=$R/gpurun_out/round3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_token.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 12 > $O/decode_step_trace.txt
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8 > $O/pmc_fetch.log 2>&1
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
python $R/tools/gemv_bench.py --fused --types q4_k,q4_0,q4_1,q8_0 > $O/gemv_fused.txt 2>&1
python $R/tools/gemv_phase_probe.py > $O/gemv_phases.txt 2>&1
for n in 1008 4080 16368; do python $R/bench.py --n-prompt $n --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|n_ctx_end": [0-9]*' | tr '\n' ' '; echo; done > $O/decode_long_context.txt
tail -1 $O/bench.json | cut -c1-400; head -8 $O/bench_kernel_stats.csv | cut -c1-150
