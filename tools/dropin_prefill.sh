#!/bin/bash
# BASELINE cfg3 through the UNMODIFIED reference host on our ggml module: Llama-3-8B shapes, Q4_0, one 4096-token prompt as a single graph
# (batch size 4096), prompt evaluation only.  Run on the GPU box.   NGL=cpu THREADS=32: the host's own CPU backend (minutes).
set -u
R=/root/repo; M=/tmp/llama3-8b-${WTYPE:-q4_0}-l4608.bin
[ -s $M ] || python $R/tools/make_ggmm.py --config llama3-8b --wtype ${WTYPE:-q4_0} --max-len 4608 --fast --out $M > /dev/null || exit 1
IDS=$(python -c "print(' '.join(str((7 * i + 11) % 32000) for i in range(${NPROMPT:-4096})))")
cd $R/oracle/_ref
CLLM_HIP_STATS=1 REF_CHAT_PREFILL_REPS=${REPS:-3} ./ref_chat $M ${NGL:-all} ${THREADS:-16} 1 - $IDS > /dev/null 2> /tmp/pf_err.txt; echo "rc=$?"
grep "^prefill:" /tmp/pf_err.txt; grep "calls (" /tmp/pf_err.txt | head -2; grep -i "error\|fail" /tmp/pf_err.txt | head -5
