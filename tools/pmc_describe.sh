#!/bin/bash
# usage (GPU box, repo root): tools/pmc_describe.sh <outdir-under-gpurun_out>   [LIBS="A B"]
# texture-addresser / L1 counters of the extractor's kernels for one or two builds of the library (lib_A.so / lib_B.so), one context.
# A variant is selected with ORBX_LIB (orb_slam2_e_amd/_lib.py), as tools/match_variants_trace.sh does: the product .so is never
# overwritten, so an interrupted run cannot leave an experiment build in its place.
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for v in ${LIBS:-B}; do
  [ -f orb_slam2_e_amd/lib_$v.so ] || { echo "no orb_slam2_e_amd/lib_$v.so"; exit 1; }
  OUT=$GRAFT_REPO_ROOT/gpurun_out/$1/$v; mkdir -p $OUT
  A="--steps 10 --pipeline 1 --no-fem --no-cpu-baseline --no-host-io --no-verify"
  run() { name=$1; shift; ( cd /tmp && export TMPDIR=/tmp ORBX_LIB=$GRAFT_REPO_ROOT/orb_slam2_e_amd/lib_$v.so && timeout -k 10 240 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py $A > $OUT/$name.log 2>&1 ); echo "$v $name rc $?"; }
  run pmc6 --pmc TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE
  run pmc7 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum
  run pmc2 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_WAVES
  python3 - "$OUT" "$v" <<'PY'
import sys, glob, csv, collections, re
out, v = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r'(k_[a-z_0-9]+)', r['Kernel_Name'])
        if not m: continue
        g = int(r.get('Grid_Size', 0) or 0)
        if g < 64 * 64 * 64: continue          # the 64-frame launches only
        acc[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
for k in ("k_describe", "k_blur", "k_fast_cells"):
    c = {n: sum(x) / len(x) for n, x in acc[k].items()}
    if not c: continue
    clocks = c.get("GRBM_GUI_ACTIVE", 0) / 8
    vm = c.get("SQ_INSTS_VMEM_RD", 0) + c.get("SQ_INSTS_VMEM_WR", 0)
    print(v, k, "TA busy %.3f" % (c.get("TA_TA_BUSY_sum", 0) / (256 * clocks) if clocks else 0),
          "L1 accesses per VMEM instr %.1f" % (c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / vm if vm else 0),
          "L1 accesses %.3g" % c.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0), "VMEM instr %.3g" % vm, "VALU instr %.3g" % c.get("SQ_INSTS_VALU", 0))
PY
done
