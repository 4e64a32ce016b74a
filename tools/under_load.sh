#!/bin/bash
# GPU box, repo root: the GPU test suite while a second process keeps the chip full (tools/saturate.py) -- two processes on the card.
# usage: tools/under_load.sh <outfile>
OUT=${1:-gpurun_out/under_load.log}
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
python3 tools/saturate.py 150 > gpurun_out/saturate.log 2>&1 &
SAT=$!
sleep 8
timeout -k 10 600 python3 -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -6 | tee $OUT
kill $SAT 2>/dev/null; wait $SAT 2>/dev/null
tail -1 gpurun_out/saturate.log | tee -a $OUT
