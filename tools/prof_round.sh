#!/bin/bash
# Evidence for profiles/ (run on the GPU box through gpurun): kernel-trace stats of the default bench command, the bench line itself,
# and the two PMC passes (FETCH_SIZE, WRITE_SIZE -- separate runs, kernel-trace only) on the dominant kernel.
set -u
R=/root/repo; O=$R/gpurun_out/round; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/bench.py --steps 48 --warmup 8 --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err   # (rocprofv3 segfaults on >= 144 graph replays on this image)
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
python $R/tools/attn_phase_probe.py 128 > $O/attn_phases.txt 2>&1
python $R/tools/prefill_bench.py --reps 2 > $O/prefill.txt 2>&1
python $R/tools/prefill_bench.py --reps 2 --wtype q4_k >> $O/prefill.txt 2>&1
for n in 1008 4080 16368; do python $R/bench.py --n-prompt $n --steps 64 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*\|n_ctx_end": [0-9]*' | tr '\n' ' '; echo; done > $O/decode_long_context.txt
rm -rf /tmp/p4; rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --kernel-trace --output-format csv -d /tmp/p4 -- python $R/tools/gemv_bench.py --types q4_0,q4_k --cols 4096 --iters 4 --shapes gate_up > $O/pmc_mfma.log 2>&1
cp $(find /tmp/p4 -name "*counter_collection.csv" | head -1) $O/pmc_mfma_mmq.csv
python $R/tools/gemv_bench.py --types q4_0,q4_1,q4_k,q8_0 --cols 4096 --iters 8 --shapes qkv,o,gate_up,down > $O/mmq_4096cols.txt 2>&1
# the drop-in path: the unmodified reference host on our module (fusion ladder, wall-time breakdown), and the runner with a host in the loop
( export NS=272
  for e in "CLLM_HIP_NO_FUSE=1" "CLLM_HIP_FUSE_ATTN=0 CLLM_HIP_PACK=0" "CLLM_HIP_FUSE_ATTN=1 CLLM_HIP_PACK=0" "CLLM_HIP_PACK=0" "CLLM_HIP_GRAPH=1" ""; do
    echo "== env: ${e:-default}"; env $e bash $R/tools/dropin_bench.sh 2>&1 | grep "^decode"
  done
  echo "== CLLM_HIP_STATS=1 (default)"; CLLM_HIP_STATS=1 bash $R/tools/dropin_bench.sh > /dev/null 2>&1; grep "per graph" /tmp/err_272.txt | tail -1; grep "calls (" /tmp/err_272.txt | tail -1
  echo "== reference host on its CPU backend"; NGL=cpu NS=24 THREADS=64 bash $R/tools/dropin_bench.sh 2>&1 | grep "^decode"
) > $O/dropin.txt 2>&1
python $R/tools/step_latency.py > $O/step_latency.txt 2>&1
for t in q4_0 q4_1 q8_0; do python $R/bench.py --wtype $t --steps 128 --warmup 8 --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*' | head -1 | sed "s/^/$t decode tok\/s /"; done > $O/decode_other_types.txt
tail -3 $O/bench.json; cat $O/decode_step_trace.txt | head -16
