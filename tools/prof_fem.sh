#!/bin/bash
# usage (GPU box, repo root): tools/prof_fem.sh <outdir-under-gpurun_out>
# HBM counters of the batched CG legs of bench.py, one leg per process so that a kernel name means one workload:
#   batch (256 x 6,591 dofs, one topology)   batch_beyond_infinity_cache (256 x 12,288 dofs)   batch_distinct_topologies
# FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md); tools/fem_traffic.py turns them into profiles/rNN_fem_traffic.json
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
for leg in batch:fem_cg_time.py:256:200:12 batch_beyond_infinity_cache:fem_cg_time.py:256:200:15 batch_distinct_topologies:fem_cg_time_distinct.py:256:200; do
  IFS=: read name script a b c <<< "$leg"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 240 rocprofv3 --pmc $ctr --output-format csv -d $OUT/$name/$ctr -- python3 $GRAFT_REPO_ROOT/tools/$script $a $b $c > $OUT/$name.$ctr.log 2>&1; echo "$name $ctr rc $?"
  done
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/trace -- python3 $GRAFT_REPO_ROOT/tools/$script $a $b $c > $OUT/$name.trace.log 2>&1; echo "$name trace rc $?"
done
