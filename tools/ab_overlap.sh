#!/bin/bash
# A/B of how the concurrent step's consumers share the chip (bench.py): "split label_on score_lds_pad lib" per line; GPU box
for cfg in "0.75 chain 0 -" "0 chain 0 prio3" "0 chain 16384 prio3" "0 chain 8192 prio3" "0.75 chain 0 prio3" "0 chain 0 -" "0 chain 12288 prio3" "0.75 chain 0 -"; do
  set -- $cfg
  lib=""; [ "$4" != "-" ] && lib="$(pwd)/rescan_amd/librescan_hip_$4.so"
  RS_HIP_LIB=$lib RS_HIP_SCORE_LDS_PAD=$3 RS_BENCH_CU_SPLIT=$1 RS_BENCH_LABEL_ON=$2 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('split $1 label on $2 score LDS pad $3 lib $4  ms/step %.3f  nn_icp %.2f icp_moments %.2f nn_score %.2f nn_label %.2f' % (d['ms_per_step'],k['nn_icp'],k['icp_moments'],k['nn_score'],k['nn_label']))"
done
