#!/bin/bash
# usage (GPU box, repo root): tools/prof_timed.sh <outdir-under-gpurun_out>
# A kernel trace whose averages ARE the timed region's: bench.py with 2,000 steps and nothing else on the card (no FEM / matcher / stereo legs,
# no host-I/O leg, no CPU baseline), so that nine of ten launches of every extract + match kernel belong to the pipelined three-context steps.
# bench.py's roofline (HIP events of the dominant kernel's own start and end, timed region only) must agree with this summary's average.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_timed -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2000 --warmup 5 --no-fem --no-cpu-baseline --no-host-io > $OUT/trace_timed.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete
tail -c 300 $OUT/trace_timed.log
