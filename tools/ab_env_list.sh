#!/bin/bash
# bash tools/ab_env_list.sh reps "<VAR=a ...>" "<VAR=b ...>" ...: the default bench (--no-extras) under each environment in turn, reps rounds
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
reps=$1; shift
for r in $(seq $reps); do
  for cfg in "$@"; do
    env $cfg python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$cfg:', round(d['ms_per_step'],4), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'pose', d['parity']['pose_dist'])"
  done
done
