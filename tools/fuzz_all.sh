#!/bin/bash
# All ten differential fuzzers against the oracle, one line each (GPU box, repo root): tools/fuzz_all.sh <outfile> [seed0]
# Counts sized for ~10 minutes in all; every fuzzer under its own timeout.
OUT=${1:-gpurun_out/fuzz_all.log}; S=${2:-341}
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
: > $OUT
run() { name=$1; n=$2; seed=$3; r=$(timeout -k 10 ${4:-300} python tests/$name.py $n $seed 2>&1 | grep -v amdgpu.ids | tail -1); echo "== $name $n $seed: $r" | tee -a $OUT; }
run fuzz_extract 600 $((S+0))
run fuzz_allpairs 900 $((S+1))
run fuzz_match 600 $((S+2))
run fuzz_search 1500 $((S+3))
run fuzz_whole 1500 $((S+4))
run fuzz_frames 6000 $((S+5))
run fuzz_loops 400 $((S+6))
run fuzz_fem_cg 200 $((S+7)) 400
run fuzz_fem_stereo 150 $((S+8))
run fuzz_fem_xcd 250 $((S+9))
