#!/bin/bash
# Round-2 evidence for profiles/ (run on the GPU box through gpurun): kernel-trace stats of the bench command, the bench line itself, the two PMC
# passes on the dominant kernels (separate runs, kernel-trace only), mat-vec timings per type, prefill by mode, long-context decode, exactness probes.
set -u
R=$PWD; O=$R/gpurun_out/round_r2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python $R/tools/trace_token.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) 14 > $O/decode_step_trace.txt
python $R/bench.py > $O/bench.json 2> $O/bench.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p2 -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8 > $O/pmc_fetch.log 2>&1
cp $(find /tmp/p2 -name "*counter_collection.csv" | head -1) $O/pmc_fetch_size.csv
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p3 -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8 > $O/pmc_write.log 2>&1
cp $(find /tmp/p3 -name "*counter_collection.csv" | head -1) $O/pmc_write_size.csv
python $R/tools/pmc_summary.py $O/pmc_fetch_size.csv $O/pmc_write_size.csv $O/pmc_summary.json > /dev/null 2>&1
python $R/tools/gemv_bench.py --fused --types q4_k,q4_0,q4_1,q8_0 > $O/gemv_fused.txt 2>&1
( python $R/tools/prefill_bench.py --reps 3; python $R/tools/prefill_bench.py --reps 3 --wtype q4_k; CLLM_PREFILL=f16 python $R/tools/prefill_bench.py --reps 3; CLLM_PREFILL=f16 python $R/tools/prefill_bench.py --reps 3 --wtype q4_k ) 2>&1 | grep "^prefill" > $O/prefill.txt
python $R/tools/prefill_modes_probe.py > $O/prefill_modes.txt 2>&1
for n in 1008 4080 16368; do python $R/bench.py --n-prompt $n --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|n_ctx_end": [0-9]*' | tr '\n' ' '; echo; done > $O/decode_long_context.txt
for t in q4_0 q4_1 q8_0; do python $R/bench.py --wtype $t --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$t decode tok\/s /"; done > $O/decode_other_types.txt
python $R/tools/exact_probe.py > $O/op_exactness_probe.txt 2>&1
python $R/tools/e2e_probe.py --big 2>&1 | cut -c1-260 > $O/e2e_exactness_probe.txt
for n in 80 300 544; do python $R/tools/attn_phase_probe.py $n 2>/dev/null | grep -v "^\[rank"; done > $O/attn_phases.txt
python $R/tools/fattn_bench.py > $O/flash_attention.txt 2>&1
python $R/tools/kq_bench.py > $O/k_quants_gemv.txt 2>&1
( cd $R && LAYERS=4 bash tools/prof_prefill.sh round_r2_prefill > /dev/null 2>&1; cp $R/gpurun_out/round_r2_prefill/prefill_kernel_stats.csv $O/prefill_kernel_stats.csv )
( cd $R && bash tools/pmc_kernel.sh "k_gemv_dec<12, 1, 1, 1" round_r2_pmc -- python $R/tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu --iters 8 > /dev/null 2>&1; cp $R/gpurun_out/round_r2_pmc/pmc.txt $O/pmc_sq_gate_up.txt )
( NS="144 400 144" bash $R/tools/dropin_bench.sh; echo "--- -fa 1 (FLASH_ATTN_EXT on the module) ---"; REF_CHAT_FA=1 NS="144 400" bash $R/tools/dropin_bench.sh; echo "--- -fa 1 --cache_dtype q8_0 ---"; REF_CHAT_FA=1 REF_CHAT_CACHE=q8_0 NS="144" bash $R/tools/dropin_bench.sh ) > $O/dropin_reference_host.txt 2>&1
( bash $R/tools/dropin_mixtral.sh ) > $O/dropin_mixtral.txt 2>&1
$R/oracle/_ref/ref_backend_async $R/oracle/_ref/libggml-hip.so 4194304 > $O/backend_async_events.txt 2>&1
tail -c 600 $O/bench.json; cat $O/decode_step_trace.txt | head -12; cat $O/pmc_summary.json | head -20
