#!/bin/bash
# Interleaved A/B of the concurrent step (bench.py, 30 steps) over environment settings and library variants; GPU box.
#   usage: bash tools/ab_overlap.sh "<split> <label_on> <score_lds_pad> <lib-tag|->" ...      (one quoted configuration per argument)
#   split = RS_BENCH_CU_SPLIT (0: no CU partition), label_on = RS_BENCH_LABEL_ON, score_lds_pad = RS_HIP_SCORE_LDS_PAD,
#   lib-tag = a library built by tools/variant.sh <tag> -D... ("-": the default build).  profiles/r03/ab_*.txt were made with it.
[ $# -eq 0 ] && set -- "0.75 chain 0 -" "0.625 batch 0 -" "0.75 chain 0 -" "0.625 batch 0 -"
for cfg in "$@"; do
  set -- $cfg
  if [ "$4" != "-" ]; then export RS_HIP_LIB="$(pwd)/rescan_amd/librescan_hip_$4.so"; else unset RS_HIP_LIB; fi
  RS_HIP_SCORE_LDS_PAD=$3 RS_BENCH_CU_SPLIT=$1 RS_BENCH_LABEL_ON=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('split $1 label on $2 score LDS pad $3 lib $4  ms/step %.3f  nn_icp %.2f icp_moments %.2f nn_score %.2f nn_label %.2f' % (d['ms_per_step'],k['nn_icp'],k['icp_moments'],k['nn_score'],k['nn_label']))"
done
