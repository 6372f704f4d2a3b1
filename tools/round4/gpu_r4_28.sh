#!/bin/bash
# round 4, call 28: calibrate SQ_VALU_MFMA_BUSY_CYCLES on the microbenchmark (known MFMA count), and count the MFMA instructions of the attention kernels
O=gpurun_out/r4_28; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_micro -- $GRAFT_REPO_ROOT/tools/micro/bin/mfma_f32_occupancy > $GRAFT_REPO_ROOT/$O/micro.log 2>&1
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -- python $GRAFT_REPO_ROOT/tools/prefill_bench.py --reps 1 --layers 2 > $GRAFT_REPO_ROOT/$O/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' | tee $O/mfma_counter_calibration.txt
import csv, glob, collections
for d, pick in (("pmc_micro", lambda k: "k_chain" in k), ("pmc", lambda k: "k_mmf_exact" in k or "k_mmx" in k)):
    fs = glob.glob("gpurun_out/r4_28/%s/**/*counter_collection.csv" % d, recursive=True)
    if not fs: print("no counter file in", d); continue
    rows = list(csv.DictReader(open(fs[0])))
    if d == "pmc_micro":
        byd = collections.defaultdict(dict)
        for r in rows:
            if pick(r["Kernel_Name"]): byd[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]] = float(r["Counter_Value"])
        for (did, k, g), cs in sorted(byd.items(), key=lambda t: int(t[0][0])):
            print("%-22s grid %-8s " % (k, g) + "  ".join("%s=%.0f" % (c, v) for c, v in sorted(cs.items())))
    else:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            if pick(r["Kernel_Name"]): acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(k)
            for c, v in sorted(cs.items()):
                v2 = v[len(v) // 2:]
                print("   %-30s %14.0f" % (c, sum(v2) / len(v2)))
PY
rm -rf $O/pmc $O/pmc_micro
