#!/bin/bash
O=gpurun_out/r3ah; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_fattn.py tests/test_gpu_llama.py -q -x -k "f16 or mul_mat_f or attn_prefill or long_prompt or attention" 2>&1 | tail -3 | tee $O/pytest.txt
timeout 300 python tools/prefill_bench.py --reps 3 2>&1 | tail -1 | tee $O/prefill_exact.txt
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_e
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -o p -- python /root/repo/tools/prefill_bench.py --layers 4 --reps 2 > /dev/null 2>&1
f=$(find /tmp/prof_e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" | cut -d, -f1-4 | sed 's/(.*)"/"/' | cut -c1-120 | tee -a /root/repo/$O/prefill_exact.txt
