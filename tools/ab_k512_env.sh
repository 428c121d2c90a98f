#!/bin/bash
# bash tools/ab_k512_env.sh "<VAR=a>" "<VAR=b>": the K = 512 strong batch under two environments
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
for r in 1 2; do
  for cfg in "$1" "$2"; do
    env $cfg python bench.py --no-cpu-baseline --no-extras --scaling strong --strong-problems 512 --steps 5 2>/dev/null > /tmp/ab.json
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$cfg:', round(d['ms_per_step'],3), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'pose', d['parity']['pose_dist'])"
  done
done
