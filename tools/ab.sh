#!/bin/bash
# A/B of two builds of the library on the same GPU box (box-to-box spread is larger than most kernel changes):
# tools/ab.sh <rounds>; expects orb_slam2_e_amd/lib_A.so and lib_B.so, alternates them under bench.py.
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-3}); do
  for v in ${LIBS:-A B}; do
    cp orb_slam2_e_amd/lib_$v.so orb_slam2_e_amd/liborbslam_hip.so || exit 1
    timeout -k 10 300 python bench.py --no-fem --no-cpu-baseline --no-host-io --no-verify --steps ${AB_STEPS:-200} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'],4), 'verified', d['verified'], {k: round(x,4) for k,x in d['kernel_ms_per_step_untimed_pass'].items()})" || exit 1
  done
done
cp orb_slam2_e_amd/lib_B.so orb_slam2_e_amd/liborbslam_hip.so
