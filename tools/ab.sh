#!/bin/bash
# run the bench (serial and concurrent) for the default build and every variant given: tools/ab.sh tag1 tag2 ...
cd "$(dirname "$0")/.."
one() { python bench.py --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', '$2', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernel_ms_per_step'].items()})"; }
for t in default "$@"; do
  if [ $t = default ]; then unset RS_HIP_LIB; else export RS_HIP_LIB=$PWD/rescan_amd/librescan_hip_$t.so; fi
  one $t --serial; one $t ""
done
