#!/bin/bash
# Development aid: each kernel's cost in the pipelined step.  tools/skip_hook.patch (never part of the product build) adds
# an ORBX_SKIP bit mask to the launch sequence: 1 level0, 2 resize, 4 FAST, 8 octree, 16 blur, 32 describe, 64 match.
# Locally:   git apply tools/skip_hook.patch && make -C orb_slam2_e_amd/csrc && cp orb_slam2_e_amd/liborbslam_hip.so \
#            orb_slam2_e_amd/lib_skip.so && git apply -R tools/skip_hook.patch && make -C orb_slam2_e_amd/csrc
# On the GPU box: tools/saturated.sh  -> ms per 64-frame step with only ONE kernel kind running on the 3 streams
# ("saturated" time; their sum is within a few % of the real step), and with everything.
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
cp orb_slam2_e_amd/liborbslam_hip.so /tmp/lib_keep.so; cp orb_slam2_e_amd/lib_skip.so orb_slam2_e_amd/liborbslam_hip.so
for s in 0 127 126 125 123 119 111 95 63 0; do
ORBX_SKIP=$s timeout -k 10 300 python bench.py --no-fem --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); n={0:'all',127:'none',126:'level0',125:'resize',123:'fast',119:'octree',111:'blur',95:'describe',63:'match'}[$s]; print('only %-9s %.4f ms/step' % (n, d['ms_per_step']))" || break
done
cp /tmp/lib_keep.so orb_slam2_e_amd/liborbslam_hip.so
