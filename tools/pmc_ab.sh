#!/bin/bash
# usage (GPU box, repo root): [PMC="counter ..."] [LIBS="A B"] tools/pmc_ab.sh <kernel-name-substring>
# counters of one kernel's 64-frame launches (one context, no overlap) for orb_slam2_e_amd/lib_A.so vs lib_B.so
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
K=${1:-k_fast_cells}
for v in ${LIBS:-A B}; do
  cp orb_slam2_e_amd/lib_$v.so orb_slam2_e_amd/liborbslam_hip.so || exit 1
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab/$v; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS} --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-fem --no-cpu-baseline --no-host-io --no-verify --steps 4 --warmup 1 --pipeline 1 > $OUT/log.txt 2>&1 ) || { tail -5 $OUT/log.txt; exit 1; }
  python3 - "$OUT" "$K" "$v" <<'PY'
import sys, glob, csv, collections
out, k, v = sys.argv[1:4]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name'] and int(r["Grid_Size"]) > int(__import__("os").environ.get("MINGRID", "1000000")):
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print(v, {c: round(a[0] / a[1]) for c, a in sorted(acc.items())}, 'launches', max(a[1] for a in acc.values()) if acc else 0)
PY
done
cp orb_slam2_e_amd/lib_B.so orb_slam2_e_amd/liborbslam_hip.so
