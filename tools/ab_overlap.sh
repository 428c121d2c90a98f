#!/bin/bash
# A/B of CU partitions for the concurrent step (bench.py): "split batch_lo label_on" per line; run on the GPU box through gpurun
for cfg in "0.75 0.75 chain" "0.625 0.625 batch" "0.625 0.625 chain" "0.625 0.625 all" "0.75 0.75 all" "0.625 0.625 batch" "0.75 0.75 chain"; do
  set -- $cfg
  RS_BENCH_CU_SPLIT=$1 RS_BENCH_CU_BATCH_LO=$2 RS_BENCH_LABEL_ON=$3 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('split $1 batch_lo $2 label on $3  ms/step %.3f  nn_icp %.2f icp_moments %.2f nn_score %.2f nn_label %.2f' % (d['ms_per_step'],k['nn_icp'],k['icp_moments'],k['nn_score'],k['nn_label']))"
done
