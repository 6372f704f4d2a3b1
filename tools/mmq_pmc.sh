cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU" "SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_MISC" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_VMEM_RD"; do
  rm -rf /tmp/pm; rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pm -- python /root/repo/tools/gemv_bench.py --types q4_0 --cols 4096 --iters 2 --shapes gate_up > /tmp/pm.log 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pm/**/*counter_collection.csv",recursive=True)
if not f: print("no output for $set"); raise SystemExit
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if "mmq" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items(): print(k, sum(v)/len(v))
PY
done
