#!/bin/bash
# usage (GPU box, repo root): tools/trace_ab.sh   -- kernel durations (rocprofv3 kernel trace, one context: no overlap) for lib_A.so vs lib_B.so
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in ${LIBS:-A B}; do
  cp orb_slam2_e_amd/lib_$v.so orb_slam2_e_amd/liborbslam_hip.so || exit 1
  OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_ab/$v; rm -rf $OUT; mkdir -p $OUT
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --no-fem --no-cpu-baseline --no-host-io --no-verify --steps 10 --warmup 2 --pipeline 1 > $OUT/log.txt 2>&1 ) || { tail -5 $OUT/log.txt; exit 1; }
  python3 - "$OUT" "$v" <<'PY'
import sys, glob, csv, collections, statistics
out, v = sys.argv[1:3]
acc = collections.defaultdict(list)
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        import re as _re
        m = _re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        if not m: continue
        n = m.group(1)
        g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
        if g >= 64 * 64:   # the 64-frame launches only
            acc[n].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0)
print(v, {n: round(statistics.median(x), 1) for n, x in sorted(acc.items()) if len(x) >= 5}, "match launches", sorted(round(t) for t in acc.get("k_match_sets_mfma", [])))
PY
done
cp orb_slam2_e_amd/lib_B.so orb_slam2_e_amd/liborbslam_hip.so
