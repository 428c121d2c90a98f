#!/bin/bash
# interleaved repeats of the concurrent bench for the default build and the variants given (box noise is a few per cent, so
# single runs do not separate variants): tools/ab_repeat.sh <repeats> tag1 tag2 ...
cd "$(dirname "$0")/.."
reps=$1; shift
for r in $(seq 1 $reps); do
  for t in default "$@"; do
    if [ $t = default ]; then unset RS_HIP_LIB; else export RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_$t.so; fi
    python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', 'mean', round(d['ms_per_step'],3), 'median', round(d['ms_per_step_spread']['median'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"
  done
done
