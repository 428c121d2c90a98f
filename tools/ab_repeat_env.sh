#!/bin/bash
# like ab_repeat.sh, with an environment setting per entry: tools/ab_repeat_env.sh <repeats> "tag|ENV=1 ENV2=2" ...   (tag = default or a variant build)
cd "$(dirname "$0")/.."
reps=$1; shift
for r in $(seq 1 $reps); do
  for e in "$@"; do
    t=${e%%|*}; envs=${e#*|}
    if [ "$t" = default ]; then lib=""; else lib="RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_$t.so"; fi
    env $lib $envs python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$e]', 'mean', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_spread']['median'],3), 'pose', '%.3e' % d['parity']['pose_dist'], {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
