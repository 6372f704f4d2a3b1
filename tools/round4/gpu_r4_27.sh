#!/bin/bash
# round 4, call 27: SQ counters of the prompt attention's kernels (one --pmc pass, kernel trace only): where do the waves' cycles go?
O=gpurun_out/r4_27; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 2 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc2 -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 2 >> $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/attention_sq_counters.txt
import csv, glob, collections
for d in ("pmc", "pmc2"):
    fs = glob.glob("gpurun_out/r4_27/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print("no counter file in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        if "k_mmf_exact" in k or "k_mmx" in k or "soft_max" in k:
            acc[k.split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            v2 = v[len(v) // 2:]
            print("   %-28s %14.0f  (mean of the last %d of %d launches)" % (c, sum(v2) / len(v2), len(v2), len(v)))
PY
tail -3 $O/run.log
rm -rf $O/pmc $O/pmc2
