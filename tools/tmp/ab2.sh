cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
cp orb_slam2_e_amd/lib_B.so orb_slam2_e_amd/liborbslam_hip.so
for s in 0 127 126 125 123 119 111 95 63 4 0; do
ORBX_SKIP=$s timeout -k 10 300 python bench.py --no-fem --no-cpu-baseline --steps 100 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip$s', round(d['value']), round(d['ms_per_step'],4))" || exit 1
done
cp orb_slam2_e_amd/lib_A.so orb_slam2_e_amd/liborbslam_hip.so
