#!/usr/bin/env python3
"""One steady-state token of a rocprofv3 --kernel-trace --memory-copy-trace run as a timeline: every kernel / copy with its start, duration and the gap in front of it.
usage: python tools/round6/trace_token.py DIR [marker-kernel-substring]   (the token = from one launch of the marker kernel to the next, taken near the end of the run)"""
import csv
import glob
import sys

d = sys.argv[1]
marker = sys.argv[2] if len(sys.argv) > 2 else "k_argmax"
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:90]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
idx = [i for i, e in enumerate(ev) if marker in e[2]]
if len(idx) < 12:
    print("marker not found often enough:", len(idx)); sys.exit(1)
a, b = idx[-10], idx[-9]
tok = ev[a:b + 1]
t0 = tok[0][0]
busy = 0
print(f"token window: {(tok[-1][0] - t0) / 1e3:.1f} us, {len(tok) - 1} ops")
agg = {}
prev_end = tok[0][0]
gaps = []
for s, e, n in tok[:-1]:
    gap = (s - prev_end) / 1e3
    gaps.append((gap, n))
    busy += (e - s)
    k = n.split("(")[0][:60]
    agg.setdefault(k, [0, 0.0, 0.0])
    agg[k][0] += 1; agg[k][1] += (e - s) / 1e3; agg[k][2] += max(gap, 0.0)
    prev_end = max(prev_end, e)
print(f"busy {busy / 1e3:.1f} us; idle {(tok[-1][0] - t0 - busy) / 1e3:.1f} us")
print("%-62s %5s %10s %12s" % ("op", "n", "busy us", "gap-before us"))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-62s %5d %10.1f %12.1f" % (k, v[0], v[1], v[2]))
print("largest gaps:")
for g, n in sorted(gaps, reverse=True)[:8]:
    print("  %.1f us before %s" % (g, n[:80]))
