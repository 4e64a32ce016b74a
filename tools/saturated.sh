#!/bin/bash
# Development aid: each kernel's cost in the pipelined step.  tools/skip_hook.patch (never part of the product build) adds
# an ORBX_SKIP bit mask to the launch sequence: 1 level0, 2 resize, 4 FAST, 8 octree, 16 blur, 32 describe, 64 match.
# Locally:   (python3 tools/make_skip_hook.py regenerates the patch when the sources have moved on)
#            git apply tools/skip_hook.patch && make -C orb_slam2_e_amd/csrc && cp orb_slam2_e_amd/liborbslam_hip.so \
#            orb_slam2_e_amd/lib_skip.so && git checkout orb_slam2_e_amd/csrc && make -C orb_slam2_e_amd/csrc
# On the GPU box: tools/saturated.sh  -> ms per 64-frame step (a) with everything, (b) WITHOUT one kernel kind (its marginal
# cost in the mix = all - without) and (c) with ONLY one kind on the 3 streams (its "saturated" time).
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
cp orb_slam2_e_amd/liborbslam_hip.so /tmp/lib_keep.so; cp orb_slam2_e_amd/lib_skip.so orb_slam2_e_amd/liborbslam_hip.so
run() { ORBX_SKIP=$1 timeout -k 10 300 python bench.py --no-fem --no-cpu-baseline --no-host-io --no-verify --steps ${SAT_STEPS:-400} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-18s %.4f ms/step' % ('$2', d['ms_per_step']))"; }
run 0 all || exit 1
for k in "1 level0" "2 resize" "4 fast" "8 octree" "16 blur" "32 describe" "64 match"; do set -- $k; run $1 "without $2"; done
for k in "126 level0" "125 resize" "123 fast" "119 octree" "111 blur" "95 describe" "63 match"; do set -- $k; run $1 "only $2"; done
run 0 all
cp /tmp/lib_keep.so orb_slam2_e_amd/liborbslam_hip.so
