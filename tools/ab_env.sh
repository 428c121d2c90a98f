#!/bin/bash
# bash tools/ab_env.sh VAR [reps]: interleaved A/B of the default bench with and without VAR=1 in the environment (ms per step, chain kernel ms)
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
var=$1; reps=${2:-3}
for r in $(seq $reps); do
  for v in "" "1"; do
    if [ -z "$v" ]; then tag="default  "; env -u $var python bench.py --no-cpu-baseline --steps 20 2>/dev/null > /tmp/ab.json; else tag="$var=1"; env $var=1 python bench.py --no-cpu-baseline --steps 20 2>/dev/null > /tmp/ab.json; fi
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$tag', round(d['ms_per_step'],4), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()}, 'pose', d['parity']['pose_dist'], 'traffic', d['roofline'].get('traffic'))"
  done
done
