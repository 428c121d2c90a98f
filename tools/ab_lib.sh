#!/bin/bash
# bash tools/ab_lib.sh <other librescan_hip.so> [reps]: interleaved A/B of the default bench (--no-extras) on the shipped library and another build of it
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
other=$1; reps=${2:-3}
for r in $(seq $reps); do
  for v in shipped other; do
    if [ $v = shipped ]; then env -u RS_HIP_LIB python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null > /tmp/ab.json; else RS_HIP_LIB=$other python bench.py --no-cpu-baseline --no-extras --steps 20 2>/dev/null > /tmp/ab.json; fi
    python -c "import json; d=json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1]); print('$v', round(d['ms_per_step'],4), {k: round(x,3) for k,x in d['kernel_ms_per_step'].items()}, 'pose', d['parity']['pose_dist'])"
  done
done
