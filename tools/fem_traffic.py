#!/usr/bin/env python3
"""HBM bytes per 200-iteration launch of the batched CG legs from a tools/prof_fem.sh output directory ->
profiles/rNN_fem_traffic.json, keyed by bench.py's leg names.  FETCH_SIZE / WRITE_SIZE are reported in KiB
(MI355X_MICROARCH.md); on gfx950 FETCH_SIZE shows half the bytes of wide coalesced streaming reads, so it is doubled
(the raw value is kept beside it); the dispatch with the largest counter value is the 200-iteration launch.
usage: fem_traffic.py gpurun_out/<dir> profiles/rNN_fem_traffic.json"""
import csv, glob, json, os, sys

d, out = sys.argv[1], sys.argv[2]
res = {}
for leg in ("batch", "batch_beyond_infinity_cache", "batch_distinct_topologies"):
    v = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        best = 0.0
        for f in glob.glob(os.path.join(d, leg, ctr, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_fem_cg_resident" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    best = max(best, float(r["Counter_Value"]))
        v[ctr] = best * 1024.0
    us = None
    for f in glob.glob(os.path.join(d, leg, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fem_cg_resident" in r["Name"]:
                us = float(r["MaxNs"]) / 1e3
    if v["FETCH_SIZE"]:
        res[leg] = {"hbm_bytes_per_launch": 2 * v["FETCH_SIZE"] + v["WRITE_SIZE"], "fetch_size_bytes_raw": v["FETCH_SIZE"],
                    "write_size_bytes": v["WRITE_SIZE"], "kernel": "k_fem_cg_resident", "launch_us_in_the_trace_pass": us,
                    "fetch_correction": "x2 (gfx950: FETCH_SIZE shows half the bytes of wide coalesced streaming reads; calibrated on 16-B/lane streams)",
                    "source": "tools/prof_fem.sh: 256 meshes x 200 iterations per launch, one leg per process, FETCH_SIZE and WRITE_SIZE in separate passes"}
    # k_fem_spmv alone (the 50 launches at the end of the leg's script): mean over its dispatches of the largest grid
    sp = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        by_grid = {}
        for f in glob.glob(os.path.join(d, leg, ctr, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "k_fem_spmv" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
                    by_grid.setdefault(int(r["Grid_Size"]), []).append(float(r["Counter_Value"]))
        if by_grid:
            v_ = by_grid[max(by_grid)]
            sp[ctr] = sum(v_) / len(v_) * 1024.0
    sus = None
    for f in glob.glob(os.path.join(d, leg, "trace", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_fem_spmv" in r["Name"]:
                sus = float(r["AverageNs"]) / 1e3
    if sp.get("FETCH_SIZE") and leg in res:
        res[leg]["k_fem_spmv"] = {"hbm_bytes_per_launch": 2 * sp["FETCH_SIZE"] + sp.get("WRITE_SIZE", 0.0), "fetch_size_bytes_raw": sp["FETCH_SIZE"],
                                  "write_size_bytes": sp.get("WRITE_SIZE", 0.0), "avg_launch_us_in_the_trace_pass": sus,
                                  "fetch_correction": "x2, as above", "source": res[leg]["source"] + "; k_fem_spmv: the 50 stand-alone launches behind the CG"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
