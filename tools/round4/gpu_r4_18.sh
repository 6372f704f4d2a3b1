#!/bin/bash
# round 4, call 18: is K.Q's time its 131 K empty (fully masked) workgroups?  causal skip vs every tile computed
O=gpurun_out/r4_18; mkdir -p $O
for nc in 0 1; do
  [ $nc = 1 ] && export CLLM_DEBUG_MMF_NONCAUSAL=1
  cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof$nc -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 2 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
  f=$(find $O/prof$nc -name "*kernel_trace.csv" | head -1)
  python - "$f" $nc <<'PY' | tee -a $O/kq.txt
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_mmf_exact' in r['Kernel_Name']]
for r in rows[-4:]:
    print('noncausal=%s grid %s x %s x %s: %.0f us' % (sys.argv[2], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
  rm -rf $O/prof$nc
done
