#!/usr/bin/env python3
"""HBM traffic per launch from a tools/prof.sh output directory (separate rocprofv3 --pmc FETCH_SIZE and
--pmc WRITE_SIZE passes), as MI355X_MICROARCH.md prescribes: both counters are reported in KiB; on gfx950
FETCH_SIZE shows exactly half of the bytes of a wide coalesced streaming read (16 B per lane), so it is
doubled for the kernels whose reads are such streams (k_fem_spmv); other access widths are uncalibrated and
kept raw.  SQ_INSTS_VALU (wave-level VALU instructions, whole chip) is copied alongside for the kernels that are
VALU-bound rather than HBM-bound.  Only the largest dispatches of each kernel (= the batch case) are averaged.
usage: prof_traffic.py gpurun_out/<dir> profiles/rNN_traffic.json "<source note>" """
import csv, glob, json, os, sys
from collections import defaultdict

STREAM16 = {"k_fem_spmv", "k_fem_cg_resident<false>", "k_fem_cg_resident<true>"}          # wide streaming reads: FETCH_SIZE x2 (calibrated at 16 B per lane; 12 B per lane since the block form: upper bound)


KEEP_TEMPLATE = ("k_fem_cg_resident",)   # <false> / <true> are different workloads (Ap in LDS / through the batch vectors)
LONGEST = ("k_fem_cg_resident",)         # one launch = n iterations: the dispatch with the largest counter value (bench: 200 iterations), not the mean


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    base = n.split("(")[0]
    return base if base.split("<")[0] in KEEP_TEMPLATE else base.split("<")[0]


PER_GRID = {"k_fem_spmv"}          # kernels whose launches of different grid sizes are different workloads: also keyed name@grid


def mean_largest(d, sub, counter):
    vals = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                vals[short(r["Kernel_Name"])][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
    out = {k: (max(v[max(v)]) if k.split("<")[0] in LONGEST else sum(v[max(v)]) / len(v[max(v)])) * 1024.0 for k, v in vals.items()}
    for k in PER_GRID & set(vals):
        for g, x in vals[k].items():
            if len(x) >= 5:
                out[f"{k}@{g}"] = sum(x) / len(x) * 1024.0
    return out


d, out, note = sys.argv[1], sys.argv[2], sys.argv[3]
fetch, write = mean_largest(d, "pmc3", "FETCH_SIZE"), mean_largest(d, "pmc4", "WRITE_SIZE")
valu = {k: v / 1024.0 for k, v in mean_largest(d, "pmc1", "SQ_INSTS_VALU").items()}   # a count, not KiB
res = {}
for k in sorted(fetch):
    if not k.startswith(("k_", "orb")):
        continue
    f, w = fetch[k], write.get(k, 0.0)
    x2 = k.split("@")[0] in STREAM16
    res[k] = {"hbm_bytes_per_launch": (2 * f if x2 else f) + w, "fetch_size_bytes_raw": f, "write_size_bytes": w,
              "fetch_correction": "x2 (gfx950 correction calibrated on 16-B/lane streaming loads; the block-form kernels stream 36 B per lane as 16 + 16 + 4, so this is an upper bound -- the raw value is kept beside it)" if x2 else "none (4-B/lane or byte loads: uncalibrated, raw value used)",
              "valu_wave_insts_per_launch": valu.get(k), "source": note}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"]) for k, v in res.items()}, indent=1))
