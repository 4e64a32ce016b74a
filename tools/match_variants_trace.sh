#!/bin/bash
# usage (GPU box, repo root): LIBS="v0 a1 ..." tools/match_variants_trace.sh -- k_match_sets_mfma durations (rocprofv3 kernel trace, launches
# back to back on one stream) for orb_slam2_e_amd/lib_<name>.so builds of the library (ORBX_LIB selects one)
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in $LIBS; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/match_variants/$v; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && ORBX_LIB=$GRAFT_REPO_ROOT/orb_slam2_e_amd/lib_$v.so timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tools/match_alone.py ${LAUNCHES:-100} > $OUT/log.txt 2>&1 ) || { tail -5 $OUT/log.txt; exit 1; }
  python3 - "$OUT" "$v" <<'PY'
import sys, glob, csv, statistics
out, v = sys.argv[1:3]
d = []
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_match_sets_mfma' in r['Kernel_Name']:
            d.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
d = d[5:]
print(v, "launches", len(d), "median us %.1f" % statistics.median(d), "min %.1f" % min(d), "mean %.1f" % statistics.mean(d), open(out + '/log.txt').read().strip().splitlines()[-1][-40:])
PY
done
