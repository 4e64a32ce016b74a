#!/bin/bash
# usage (GPU box, repo root): tools/prof_r05.sh <tag>      -> gpurun_out/<tag>_{main,default,full,fem,bow}
# Every profile the bench line's roofline objects are recomputed from, on the library as built:
#   main     tools/prof_main.sh     extract + match alone (one context): kernel trace + the PMC passes (FETCH_SIZE, WRITE_SIZE separately)
#   default  tools/prof_default.sh  the pipelined three-context run without the other legs: per-kernel averages as the timed region sees them
#   full     kernel trace of the driver's own command line (python3 bench.py --steps 20 --warmup 5)
#   timed    tools/prof_timed.sh    2,000 steps, no other leg: per-kernel averages OF the timed region (bench.py's roofline events agree with them)
#   fem      tools/prof_fem.sh      the three batched CG legs, one per process: k_fem_cg_resident and k_fem_spmv, trace + FETCH_SIZE + WRITE_SIZE
#   bow      tools/bow_transform_prof.py: k_bow_transform, trace + FETCH_SIZE + WRITE_SIZE
# Locally afterwards (tools/prof_r05_collect.sh <tag> <round>): summaries into profiles/.
T=${1:-r05}
R=$GRAFT_REPO_ROOT
$R/tools/prof_main.sh ${T}_main
$R/tools/prof_default.sh ${T}_default
find $R/gpurun_out/${T}_default -name "*kernel_trace.csv" -delete      # (the per-dispatch rows: only the stats are kept; gpurun returns 64 MiB at most)
OUT=$R/gpurun_out/${T}_full; mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/trace.err; echo "full rc $?" )
find $OUT -name "*kernel_trace.csv" -delete                             # (70,000 dispatches of the all-meshes FEM leg)
$R/tools/prof_timed.sh ${T}_timed    # 2,000 pipelined steps and nothing else: the averages the bench line's event-timed roofline must agree with
$R/tools/prof_fem.sh ${T}_fem
OUT=$R/gpurun_out/${T}_bow; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bow_transform_prof.py > $OUT/trace.log 2>&1; echo "bow trace rc $?"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$ctr -- python3 $R/tools/bow_transform_prof.py > $OUT/$ctr.log 2>&1; echo "bow $ctr rc $?"
done
