#!/bin/bash
# GPU box, repo root: the two CG fuzzers over and over with fresh seeds for about <minutes> (default 12).  usage: tools/soak_fem.sh <outfile> [minutes] [seed0]
OUT=${1:-gpurun_out/soak_fem.log}; MIN=${2:-12}; S=${3:-90000}
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
: > $OUT
END=$(( $(date +%s) + 60 * MIN ))
n=0
while [ $(date +%s) -lt $END ]; do
  for f in fuzz_fem_cg:200 fuzz_fem_xcd:250; do
    name=${f%%:*}; cnt=${f##*:}
    r=$(timeout -k 10 400 python tests/$name.py $cnt $((S + n)) 2>&1 | grep -v amdgpu.ids | grep -E "MISMATCH|cases" | tail -4 | tr '\n' '|')
    echo "== $name $cnt $((S + n)): $r" >> $OUT
  done
  n=$((n + 1)); [ $((n % 4)) -eq 0 ] && echo "pair $n done at $(date +%H:%M:%S)"
done
echo "pairs $n; lines with a mismatch count other than 0:" | tee -a $OUT
grep -E "mismatches [1-9]" $OUT | tee -a $OUT | head
