#!/usr/bin/env python3
"""Per-kernel counters of the 64-frame launches of a tools/prof_main.sh output directory: for every kernel the dispatches
with its LARGEST grid are the bench's batch launches (a level-by-level kernel like k_pyr_resize has one grid per level:
all its dispatches count, summed per step = per 7 launches).  Writes a markdown table (mean per launch) and, next to it,
derived figures: VALU issue fraction, texture-addresser busy fraction, L1 accesses per wave-wide memory instruction.
usage: prof_pick.py gpurun_out/<dir> profiles/rNN_main_pmc_summary.md"""
import csv, glob, os, sys
from collections import defaultdict

d, out = sys.argv[1], sys.argv[2]
PER_LEVEL = {"k_pyr_resize"}


def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


dur = {}
for f in glob.glob(os.path.join(d, "trace", "**", "*kernel_trace.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])].append(
            (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, g in acc.items():
        if k in PER_LEVEL:
            v = [x for xs in g.values() for x in xs]
            dur[k] = (sum(v) / len(v), len(v), "all grids")
        else:
            v = g[max(g)]
            dur[k] = (sum(v) / len(v), len(v), max(g))
ctr = defaultdict(dict)
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(list)))
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, g in cs.items():
            v = [x for xs in g.values() for x in xs] if k in PER_LEVEL else g[max(g)]
            ctr[k][c] = sum(v) / len(v)
names = sorted({c for k in ctr for c in ctr[k]})
lines = ["| kernel | launches | grid (threads) | avg us | " + " | ".join(names) + " |", "|---|---|---|---|" + "---|" * len(names)]
keep = [k for k in dur if k.startswith("k_")]
for k in sorted(keep, key=lambda k: -dur[k][0] * (7 if k in PER_LEVEL else 1)):
    lines.append(f"| {k} | {dur[k][1]} | {dur[k][2]} | {dur[k][0]:.1f} | " + " | ".join(f"{ctr[k][c]:.4g}" if c in ctr[k] else "" for c in names) + " |")
lines += ["", "Derived (per launch; 1024 SIMDs, 256 CUs; clocks = GRBM_GUI_ACTIVE / 8, the counter being summed over the 8 XCDs):", "",
          "| kernel | VALU issue fraction (SQ_INSTS_VALU x 4 / (1024 x clocks)) | TA busy fraction (TA_TA_BUSY_sum / (256 x clocks)) | L1 accesses per wave-wide VMEM instruction | L1 -> L2 read requests per L1 access |", "|---|---|---|---|---|"]
for k in sorted(keep, key=lambda k: -dur[k][0]):
    c = ctr[k]
    clk = c.get("GRBM_GUI_ACTIVE")
    if not clk:
        continue
    clk /= 8.0          # the counter is accumulated over the 8 XCDs (k_fast_cells: 1.61e6 for an 80-us launch = 8 x 2.5 GHz)
    vm = c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    f = lambda x: f"{x:.3f}" if x is not None else ""
    lines.append(f"| {k} | {f(c['SQ_INSTS_VALU'] * 4 / (1024 * clk)) if 'SQ_INSTS_VALU' in c else ''} | "
                 f"{f(c['TA_TA_BUSY_sum'] / (256 * clk)) if 'TA_TA_BUSY_sum' in c else ''} | "
                 f"{f(c['TCP_TOTAL_CACHE_ACCESSES_sum'] / vm) if vm and 'TCP_TOTAL_CACHE_ACCESSES_sum' in c else ''} | "
                 f"{f(c['TCP_TCC_READ_REQ_sum'] / c['TCP_TOTAL_CACHE_ACCESSES_sum']) if c.get('TCP_TOTAL_CACHE_ACCESSES_sum') and 'TCP_TCC_READ_REQ_sum' in c else ''} |")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-(len(keep) + 4):]))
