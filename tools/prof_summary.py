#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel mean duration and PMC
counters (mean per dispatch).  usage: prof_summary.py gpurun_out/<dir> [out.md]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return n.split("(")[0][:40]
rows = []
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
stats = {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])) for r in rows}
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
ctrs = sorted({c for k in pmc for c in pmc[k]})
lines.append("| kernel | calls | avg us | % | " + " | ".join(ctrs) + " |")
lines.append("|---|---|---|---|" + "---|" * len(ctrs))
for k, (calls, us, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    if pct < 0.5: continue
    vals = []
    for c in ctrs:
        v = pmc.get(k, {}).get(c)
        vals.append(f"{sum(v)/len(v):.4g}" if v else "")
    lines.append(f"| {k} | {calls} | {us:.1f} | {pct:.1f} | " + " | ".join(vals) + " |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
