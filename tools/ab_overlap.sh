#!/bin/bash
# A/B of library variants in the concurrent step (bench.py): "lib-tag" per line ("-": the default build); GPU box
for tag in - socc7 - socc7 - socc7 - socc7; do
  lib=""; [ "$tag" != "-" ] && lib="$(pwd)/rescan_amd/librescan_hip_$tag.so"
  if [ -n "$lib" ]; then export RS_HIP_LIB=$lib; else unset RS_HIP_LIB; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('lib $tag  ms/step %.3f  nn_icp %.2f icp_moments %.2f nn_score %.2f nn_label %.2f' % (d['ms_per_step'],k['nn_icp'],k['icp_moments'],k['nn_score'],k['nn_label']))"
done
