#!/bin/bash
# SQ counters of one kernel: bash tools/pmc_kernel.sh <kernel name substring> <out name> -- <command ...>
# (counter passes only: --pmc with --kernel-trace, never with the sys/hip traces)
KSUB="$1"; OUT="$2"; shift 3
R=$PWD; O=$R/gpurun_out/$OUT; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/pmc.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC" "SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -- "$@" > /tmp/pm.log 2>&1
  KSUB="$KSUB" python - <<'PY' >> $O/pmc.txt
import csv,glob,collections,os
f=glob.glob("/tmp/pm/**/*counter_collection.csv",recursive=True)
if not f:
    print("no output"); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if os.environ["KSUB"] in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, sum(v)/len(v), "n=%d" % len(v))
PY
done
cat $O/pmc.txt
