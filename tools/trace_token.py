#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV of bench.py: per-kernel averages and the launch sequence of one decode step."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000
agg = collections.defaultdict(list)
for r in rows:
    agg[(r["Kernel_Name"][:44], r["Grid_Size_X"], r["Workgroup_Size_X"])].append(dur(r))
print(f"{'kernel':46s} {'grid':>8s} {'wg':>5s} {'calls':>6s} {'avg us':>8s} {'total ms':>9s}")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{k[0]:46s} {k[1]:>8s} {k[2]:>5s} {len(v):6d} {sum(v)/len(v):8.2f} {sum(v)/1000:9.2f}")
af = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_argmax_final")]
if len(af) > 3:
    seg = rows[af[-2] + 1: af[-1] + 1]
    t0 = int(rows[af[-2]]["End_Timestamp"])
    t1 = int(rows[af[-1]]["End_Timestamp"])
    busy = sum(dur(r) for r in seg)
    print(f"last decode step: {len(seg)} launches, wall {(t1 - t0)/1000:.1f} us, sum of kernel durations {busy:.1f} us")
    prev = t0
    for r in seg[:6] + seg[-3:]:
        print(f"   {r['Kernel_Name'][:40]:40s} grid {r['Grid_Size_X']:>7s} dur {dur(r):7.2f} us  gap {(int(r['Start_Timestamp']) - prev)/1000:6.2f} us")
        prev = int(r["End_Timestamp"])
