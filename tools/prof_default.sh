#!/bin/bash
# usage (GPU box, repo root): tools/prof_default.sh <outdir-under-gpurun_out>
# kernel trace of the DEFAULT pipelined run (three contexts) without the FEM / stereo / matcher-loop legs: per-kernel averages as the timed region sees them
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_default -- python3 $GRAFT_REPO_ROOT/bench.py --no-fem --no-cpu-baseline > $OUT/trace_default.log 2>&1
tail -c 400 $OUT/trace_default.log
