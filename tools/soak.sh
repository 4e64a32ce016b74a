#!/bin/bash
# GPU box, repo root: the ten fuzzers over and over with fresh seeds for about <minutes> (default 15), then the thread stress test.
# usage: tools/soak.sh <outfile> [minutes] [seed0]
OUT=${1:-gpurun_out/soak.log}; MIN=${2:-15}; S=${3:-20000}
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
: > $OUT
END=$(( $(date +%s) + 60 * MIN ))
n=0
while [ $(date +%s) -lt $END ]; do
  bash tools/fuzz_all.sh gpurun_out/soak_pass.log $((S + 100 * n)) > /dev/null 2>&1
  cat gpurun_out/soak_pass.log >> $OUT
  n=$((n + 1)); echo "pass $n done at $(date +%H:%M:%S)"
done
timeout -k 10 120 python tests/stress_threads.py 2>&1 | grep -v amdgpu.ids | tail -2 | tee -a $OUT
echo "passes $n; lines with a mismatch or failure count other than 0:" | tee -a $OUT
grep -E "mismatches [1-9]|failures [1-9]|[1-9][0-9]* failures" $OUT | tee -a $OUT | head
