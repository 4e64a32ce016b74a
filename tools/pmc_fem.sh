#!/bin/bash
# usage (GPU box, repo root): [PMC="counter ..."] tools/pmc_fem.sh <kernel-name-substring> [fem_cg_time.py args]
# counters of one FEM kernel (sum over its dispatches / dispatches) while tools/fem_cg_time.py runs
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
K=${1:-k_fem_cg_resident}; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_fem; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc ${PMC:-SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS} --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/fem_cg_time.py "$@" > $OUT/log.txt 2>&1 ) || { tail -5 $OUT/log.txt; exit 1; }
python3 - "$OUT" "$K" <<'PY'
import sys, glob, csv, collections
out, k = sys.argv[1:3]
acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
print({c: round(a[0] / a[1]) for c, a in sorted(acc.items())}, 'dispatches', max(a[1] for a in acc.values()) if acc else 0)
PY
