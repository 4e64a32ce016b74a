#!/bin/bash
# A/B of one library under an environment switch: tools/ab_env.sh VAR rounds
cd $GRAFT_REPO_ROOT; export PYTHONPATH=$GRAFT_REPO_ROOT
for i in $(seq 1 ${2:-3}); do
  for v in 0 1; do
    env $1=$v timeout -k 10 300 python bench.py --no-fem --no-cpu-baseline --no-host-io --no-verify --steps ${AB_STEPS:-200} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1=$v', round(d['value']), round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['kernel_ms_per_step_untimed_pass'].items()})" || exit 1
  done
done
