#!/bin/bash
# usage (here, after gpurun merged gpurun_out/): tools/prof_r05_collect.sh <tag> <rNN>   -> profiles/<rNN>_*
T=$1; R=$2; G=gpurun_out; P=profiles
python3 tools/prof_pick.py $G/${T}_main $P/${R}_main_pmc_summary.md > /dev/null
python3 tools/prof_traffic.py $G/${T}_main $P/${R}_traffic.json "run ${T}: tools/prof_main.sh (bench.py --steps 10 --pipeline 1 --no-fem), separate FETCH_SIZE / WRITE_SIZE passes" > /dev/null
cp $(find $G/${T}_main/trace -name "*kernel_stats.csv" | head -1) $P/${R}_main_bench_kernel_stats.csv
cp $(find $G/${T}_default/trace_default -name "*kernel_stats.csv" | head -1) $P/${R}_default_bench_kernel_stats.csv
cp $(find $G/${T}_full/trace -name "*kernel_stats.csv" | head -1) $P/${R}_full_bench_kernel_stats.csv
tail -1 $G/${T}_full/bench_line.json > $P/${R}_bench_line_under_rocprof.json
if [ -d $G/${T}_timed ]; then cp $(find $G/${T}_timed/trace_timed -name "*kernel_stats.csv" | head -1) $P/${R}_timed_bench_kernel_stats.csv; grep '^{' $G/${T}_timed/trace_timed.log | tail -1 > $P/${R}_bench_line_timed_under_rocprof.json; fi
python3 tools/fem_traffic.py $G/${T}_fem $P/${R}_fem_traffic.json > /dev/null
for leg in batch batch_beyond_infinity_cache batch_distinct_topologies; do
  cp $(find $G/${T}_fem/$leg/trace -name "*kernel_stats.csv" | head -1) $P/${R}_fem_${leg}_kernel_stats.csv
done
cp $(find $G/${T}_bow/trace -name "*kernel_stats.csv" | head -1) $P/${R}_bow_kernel_stats.csv
python3 - "$G/${T}_bow" "$P/${R}_traffic.json" "$T" <<'PY'
import csv, glob, json, os, sys
d, out, tag = sys.argv[1:4]
v = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    x = [float(r["Counter_Value"]) for f in glob.glob(os.path.join(d, ctr, "**", "*counter_collection.csv"), recursive=True)
         for r in csv.DictReader(open(f)) if "k_bow_transform" in r["Kernel_Name"] and r["Counter_Name"] == ctr and int(r["Grid_Size"]) > 1000000]
    v[ctr] = sum(x) / max(len(x), 1) * 1024.0
t = json.load(open(out))
t["k_bow_transform"] = {"hbm_bytes_per_launch": v["FETCH_SIZE"] + v["WRITE_SIZE"], "fetch_size_bytes_raw": v["FETCH_SIZE"], "write_size_bytes": v["WRITE_SIZE"],
                        "fetch_correction": "none (32-byte gathers: uncalibrated, raw value; Infinity-Cache hits are counted as fetches)",
                        "valu_wave_insts_per_launch": None, "source": f"run {tag}: tools/bow_transform_prof.py, 64 x 2000 descriptors per launch, separate FETCH_SIZE / WRITE_SIZE passes"}
json.dump(t, open(out, "w"), indent=1)
print("k_bow_transform", t["k_bow_transform"]["hbm_bytes_per_launch"])
PY
ls -la $P/${R}_*
