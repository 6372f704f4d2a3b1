#!/usr/bin/env python3
"""profiles/r02_pmc_summary.json from the two rocprofv3 --pmc passes of tools/prof_round2.sh (FETCH_SIZE and WRITE_SIZE counter_collection CSVs).
FETCH_SIZE is reported in KB and, on gfx950, at 1/2 of the bytes of a wide coalesced stream (MI355X_MICROARCH.md, HBM section): bytes = KB * 1024 * 2.
usage: python tools/pmc_summary.py FETCH.csv WRITE.csv OUT.json"""
import collections
import csv
import json
import sys

LABEL = {
    "k_gemv_dec<12, 1, 1, 1": ("k_gemv_dec<12, 1, 1, 1> (gate/up GEMV 28672x4096, decode form)", 28672 * 4096 // 256 * 144),
    "k_gemv_rows<1, 0, 1, 8": ("k_gemv_rows<1, 0, 1, 8> (lm_head GEMV 128256x4096, LDS-staged form)", 128256 * 4096 // 256 * 144),
    "k_gemv_dec<12, 1, 0, 1": ("k_gemv_dec<12, 1, 0, 1> (lm_head GEMV 128256x4096, decode form)", 128256 * 4096 // 256 * 144),
}


def collect(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        for k in LABEL:
            if k in r["Kernel_Name"]:
                agg[k].append(float(r["Counter_Value"]))
    return agg


fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
out = {"_how": "cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/gemv_bench.py --fused --types q4_k --shapes gate_up_silu,lm_head --iters 8"
               "   (WRITE_SIZE in a second, separate pass; tools/prof_round2.sh).  FETCH_SIZE is reported in KB and, on gfx950, at 1/2 of the bytes of a wide coalesced "
               "stream (MI355X_MICROARCH.md, HBM section): bytes = FETCH_SIZE * 1024 * 2.  WRITE_SIZE is uncalibrated on gfx950 (reported as is, KB)."}
for k, (label, alg) in LABEL.items():
    if k not in fetch:
        continue
    v = fetch[k][len(fetch[k]) // 4:] or fetch[k]          # skip the warm-up launches
    kb = sum(v) / len(v)
    out[label] = {"fetch_size_kb": round(kb, 2), "hbm_read_bytes": int(kb * 1024 * 2), "algorithmic_bytes": alg, "ratio": round(kb * 2048 / alg, 4),
                  "write_size_kb_uncalibrated": round(sum(write.get(k, [0])) / max(1, len(write.get(k, [0]))), 1), "launches_averaged": len(v)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
